"""Functional stand-ins for the three things the reference's TRAINING script takes from optional packages that an
offline MI355X box does not have (torchvision, tensorboard).  `gangealing_amd.launch.stub_missing` installs them only
into modules it had to stub because the real package is absent; with the real packages installed nothing here is used.

  torchvision.models.vgg16            -> `.features` of configuration 'D' (lpips_backbones.py:101 takes only `.features`
                                         and loads the SimCLR / ImageNet weights into it)
  torchvision.utils.make_grid         -> the image grid of utils/vis_tools/helpers.py:37-41 (training visuals)
  torch.utils.tensorboard.SummaryWriter -> scalars appended to <log_dir>/scalars.jsonl (train.py:145-150)
"""
import builtins
import json
import math
import os
import types

import torch
import torch.nn as nn


def vgg16(pretrained=False, **kwargs):
    """VGG16 feature stack (configuration 'D': 13 3x3 convolutions + ReLU, five 2x2 max-pools), randomly initialised;
    `pretrained=True` cannot be honoured offline and raises."""
    if pretrained:
        raise RuntimeError('torchvision is not installed: ImageNet VGG16 weights cannot be fetched; load a '
                           '`features` state_dict explicitly')
    cfg = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']
    layers, cin = [], 3
    for v in cfg:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return types.SimpleNamespace(features=nn.Sequential(*layers))


def make_grid(tensor, nrow=8, padding=2, normalize=False, range=None, value_range=None, scale_each=False, pad_value=0.0,
              **kwargs):
    """(N, C, H, W) -> (C, rows * (H + padding) + padding, cols * (W + padding) + padding): images left to right, top to
    bottom, `nrow` per row; normalize maps [low, high] (given, or each image's / the batch's min and max) to [0, 1]."""
    if isinstance(tensor, (list, tuple)):
        tensor = torch.stack(list(tensor), 0)
    if tensor.dim() == 2:                    # a single (H, W) image
        tensor = tensor.unsqueeze(0)
    if tensor.dim() == 3:
        tensor = tensor.unsqueeze(0)
    tensor = tensor.detach().float().clone()
    if tensor.size(1) == 1:
        tensor = tensor.expand(-1, 3, -1, -1).clone()
    rng = value_range if value_range is not None else range
    if normalize:
        def norm_(t, lo, hi):
            t.clamp_(min=lo, max=hi).sub_(lo).div_(max(hi - lo, 1e-5))
        if scale_each:
            for t in tensor:
                norm_(t, *(rng if rng is not None else (float(t.min()), float(t.max()))))
        else:
            norm_(tensor, *(rng if rng is not None else (float(tensor.min()), float(tensor.max()))))
    if tensor.size(0) == 1:                  # torchvision hands a single image back as it is: no padding frame
        return tensor.squeeze(0)
    n, c, h, w = tensor.shape
    cols = min(nrow, n)
    rows = int(math.ceil(n / cols))
    grid = tensor.new_full((c, rows * (h + padding) + padding, cols * (w + padding) + padding), float(pad_value))
    for k in builtins.range(n):              # (`range` is a parameter here, as in the signature it mirrors)
        y, x = (k // cols) * (h + padding) + padding, (k % cols) * (w + padding) + padding
        grid[:, y:y + h, x:x + w] = tensor[k]
    return grid



def save_image(tensor, fp, **kwargs):
    from PIL import Image
    grid = make_grid(tensor, **kwargs)
    arr = grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to('cpu', torch.uint8).numpy()
    Image.fromarray(arr).save(fp)


class SummaryWriter:
    """Scalars as JSON lines under the log directory; every other logging call is accepted and dropped."""

    def __init__(self, log_dir=None, *args, **kwargs):
        self.log_dir = log_dir or 'runs'
        os.makedirs(self.log_dir, exist_ok=True)
        self._path = os.path.join(self.log_dir, 'scalars.jsonl')

    def add_scalar(self, tag, scalar_value, global_step=None, *args, **kwargs):
        with open(self._path, 'a') as f:
            f.write(json.dumps({'tag': tag, 'value': float(scalar_value), 'step': global_step}) + '\n')

    def flush(self):
        pass

    def close(self):
        pass

    def __getattr__(self, name):
        if name.startswith('add_'):
            return lambda *a, **k: None
        raise AttributeError(name)
