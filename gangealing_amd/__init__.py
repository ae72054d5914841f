"""gangealing_amd - MI355X-native (gfx950) implementation of the GANgealing training hot path.

Only what the path needs lives here:
  csrc/                 hand-written HIP kernels behind the C ABI in include/gangealing_hip.h
  _lib.py               ctypes binding (raw device pointers + current HIP stream)
  op/                   drop-in for models/stylegan2/op  (upfirdn2d, fused_act, conv2d_gradfix)
  splat2d_cuda/         drop-in for utils/splat2d_cuda
  spatial_transformers/ drop-in for models/spatial_transformers (anti-aliased sampling, heads, STN)
  stylegan2/            the generator built on the fused modulated convolution
  losses, latent_learner, distributed, train_step: the callers on either side of the path
  cluster_classifier    ResnetClassifier + its training iteration (clustering variants; §8 f3)
  launch.py             runs an unmodified reference script with these modules injected
"""
__version__ = '0.1.0'
