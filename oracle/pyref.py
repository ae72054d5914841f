"""Where the reference's own Python lives, and the two ways the checker imports it.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): used by tests/, bench.py's `cpu_baseline` leg and bench.py's
`extras.dropin_route`; nothing under gangealing_amd/ imports this.

The reference checkout (/root/reference) exists in the authoring container only.  `make -C oracle` stages every `*.py`
of it under the git-ignored oracle/_ref/pyref/ (oracle/Makefile), which travels to the GPU box with the snapshot the
same way oracle/_ref/libsplat_ref.so does - so the reference's modules can be driven ON the MI355X:

  hip_api()   the literal drop-in: `gangealing_amd.launch.inject(root)` pre-populates sys.modules with the HIP operator
              modules (models.stylegan2.op.*, utils.splat2d_cuda.*, antialiased_sampling), then the reference's
              networks.py / spatial_transformer.py / warping_heads.py / latent_learner.py / loss.py / lpips.py are
              imported UNMODIFIED - what `python -m gangealing_amd.launch train.py` gives train.py:89-134.
  cpu_api()   the reference on its pure-PyTorch CPU op fallback (oracle/make_golden.import_reference: JIT build
              neutralised, upfirdn2d_native / CPU fused_leaky_relu) - the fixture generator and the CPU baseline.

One process can hold only one of the two (both bind `models.*` in sys.modules); bench.py therefore runs its CPU
baseline in a child process.
"""
import os
import sys
import types

import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
STAGED = os.path.join(HERE, '_ref', 'pyref')


def find_root():
    """-> directory holding the reference's `models/`, `utils/`, `train.py`, or None.  The live checkout wins (authoring
    container); the staged copy is what the GPU box has."""
    for root in (os.environ.get('GANGEALING_REFERENCE'), '/root/reference', STAGED):
        if root and os.path.isfile(os.path.join(root, 'models', 'stylegan2', 'networks.py')):
            return root
    return None


def local_vgg16(pretrained=False, **kwargs):
    """Stand-in for torchvision.models.vgg16 (not installed): only `.features` is used (lpips_backbones.py:101); the
    layer list is torchvision's cfg 'D'."""
    cfg = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']
    layers, cin = [], 3
    for v in cfg:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return types.SimpleNamespace(features=nn.Sequential(*layers))


def _namespace():
    """The reference's public names of the training path, as imported from whatever `models` package sys.path /
    sys.modules currently resolve (train.py:13-19)."""
    if 'torchvision.models' in sys.modules and not hasattr(sys.modules['torchvision.models'], '__file__'):
        sys.modules['torchvision.models'].vgg16 = local_vgg16
    from models.stylegan2.networks import Generator
    from models.spatial_transformers.spatial_transformer import get_stn
    from models.spatial_transformers.antialiased_sampling import BilinearDownsample
    from models.latent_learner import DirectionInterpolator
    from models.losses.lpips import LPIPS
    from models.losses.loss import (gangealing_loss, gangealing_cluster_loss, total_variation_loss,
                                    flow_identity_loss)
    from models import accumulate
    import models
    return types.SimpleNamespace(Generator=Generator, get_stn=get_stn, BilinearDownsample=BilinearDownsample,
                                 DirectionInterpolator=DirectionInterpolator, LPIPS=LPIPS,
                                 gangealing_loss=gangealing_loss, gangealing_cluster_loss=gangealing_cluster_loss,
                                 total_variation_loss=total_variation_loss, flow_identity_loss=flow_identity_loss,
                                 accumulate=accumulate, root=os.path.dirname(os.path.dirname(models.__file__)))


def purge_models():
    """Forget every `models*` module so that the next import resolves them afresh (a process can hold the reference's
    model modules OR the launcher's --modules bindings, and the two routes are measured one after the other)."""
    for name in [k for k in sys.modules if k == 'models' or k.startswith('models.')]:
        del sys.modules[name]


def hip_api(root=None, modules=False):
    """The reference's modules with the HIP operators standing in for its CUDA extensions (the launcher's route);
    modules=True: the launcher's `--modules` route (this package's generator / STN / loss modules under the reference's
    names, the reference's own models/__init__.py, latent_learner.py and training glue)."""
    root = root or find_root()
    if root is None:
        return None
    from gangealing_amd import launch
    purge_models()
    launch.inject(root, modules=modules)
    return _namespace()


def cpu_api(root=None):
    """The reference's modules on its own pure-PyTorch CPU fallback."""
    root = root or find_root()
    if root is None:
        return None
    os.environ['GANGEALING_REFERENCE'] = root
    from oracle import make_golden
    make_golden.REF = root
    make_golden.import_reference()
    return _namespace()
