"""Drop-in for models/spatial_transformers (anti-aliased sampling, warp heads, STN)."""
