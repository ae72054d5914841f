"""Deterministic, name-keyed tensors.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference publishes no checkpoints we can reach offline, so model-level golden vectors use
weights generated from the parameter NAME: the same state_dict (identical key names - the
product keeps the reference's names for checkpoint compatibility) is loaded into
the reference model (by oracle/make_golden.py, authoring container only) and into ours.
"""
import zlib

import numpy as np


def det_array(name, shape, scale=1.0, dtype=np.float32):
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return (rs.standard_normal(size=tuple(shape)) * scale).astype(dtype)


def det_state_dict(module, scale_rules=()):
    """State dict for `module` (a torch.nn.Module) with every parameter drawn from det_array.
    scale_rules: sequence of (substring, scale); first match wins; default scale 1.0 for
    weights, 0.1 for biases / noise weights (so nothing is exactly zero and every path is live)."""
    import torch
    sd = {}
    for name, p in module.named_parameters():
        scale = None
        for sub, s in scale_rules:
            if sub in name:
                scale = s
                break
        if scale is None:
            scale = 0.1 if (name.endswith('bias') or 'noise' in name) else 1.0
        sd[name] = torch.from_numpy(det_array(name, p.shape, scale))
    return sd
