"""Run an UNMODIFIED reference script (e.g. wpeebles/gangealing train.py) on the MI355X operators.

    python -m gangealing_amd.launch /path/to/gangealing/train.py --exp-name cats --ckpt cat ...
    torchrun --nproc_per_node=8 -m gangealing_amd.launch /path/to/gangealing/train.py ...

Before the script is imported, the reference's operator packages are pre-populated in sys.modules
with this package's drop-ins, so `from models.stylegan2.op import ...` (networks.py:6),
`from utils.splat2d_cuda import splat2d` (helpers.py:8, spatial_transformer.py:299) and
`from models.spatial_transformers.antialiased_sampling import ...` (warping_heads.py:10,
models/__init__.py:6) resolve to the HIP implementations.  The reference's own op modules JIT-compile
CUDA at import time (upfirdn2d.py:12-18, fused_act.py:11-17) and are therefore never imported.
Optional python dependencies that are absent offline (torchvision, lmdb, tensorboard) are stubbed
only when missing.

    python -m gangealing_amd.launch --modules /path/to/gangealing/train.py ...      (or GANGEALING_LAUNCH_MODULES=1)

additionally stands this package's MODULES in for the reference's generator, spatial transformer and loss modules
(models.stylegan2.networks, models.spatial_transformers.{spatial_transformer, warping_heads}, models.losses.{loss,
lpips}): same class names, constructor arguments and state_dict layouts (reference checkpoints load), but the modulated
convolutions run in their shared-weight form instead of materialising (N*Cout, Cin, k, k) weights for a grouped
convolution (networks.py:233-282) - the script, its optimizers, schedules, checkpoints and logging stay the reference's.
Measured: bench.py `extras.dropin_modules` beside `extras.dropin_route`.
"""
import importlib
import os
import runpy
import sys
import types

OP_MODULES = {
    'models.stylegan2.op': 'gangealing_amd.op',
    'models.stylegan2.op.upfirdn2d': 'gangealing_amd.op.upfirdn2d',
    'models.stylegan2.op.fused_act': 'gangealing_amd.op.fused_act',
    'models.stylegan2.op.conv2d_gradfix': 'gangealing_amd.op.conv2d_gradfix',
    'utils.splat2d_cuda': 'gangealing_amd.splat2d_cuda',
    'utils.splat2d_cuda.functional': 'gangealing_amd.splat2d_cuda.functional',
    'utils.splat2d_cuda.splat': 'gangealing_amd.splat2d_cuda.splat',
    'models.spatial_transformers.antialiased_sampling': 'gangealing_amd.spatial_transformers.antialiased_sampling',
}


# --modules: the reference's model modules -> this package's (a module of ours, or a dict of names taken from one)
_LOSS_NAMES = ('total_variation_loss', 'flow_identity_loss', 'sample_gan_supervised_pairs', 'gangealing_loss',
               'assign_fake_images_to_clusters', 'gangealing_cluster_loss')
_LPIPS_NAMES = ('get_perceptual_loss', 'LPIPS', 'ScalingLayer', 'NetLinLayer', 'vgg16')
MODEL_MODULES = {
    'models.stylegan2.networks': 'gangealing_amd.stylegan2.networks',
    'models.spatial_transformers.spatial_transformer': 'gangealing_amd.spatial_transformers.spatial_transformer',
    'models.spatial_transformers.warping_heads': 'gangealing_amd.spatial_transformers.warping_heads',
    'models.losses.loss': ('gangealing_amd.losses', _LOSS_NAMES),
    'models.losses.lpips': ('gangealing_amd.losses', _LPIPS_NAMES),
}


class _Stub(types.ModuleType):
    def __getattr__(self, item):
        if item.startswith('__'):
            raise AttributeError(item)
        return type(item, (), {'__init__': lambda self, *a, **k: None, '__call__': lambda self, *a, **k: None})


def stub_missing(names=('torchvision', 'torchvision.models', 'torchvision.datasets', 'torchvision.datasets.utils',
                        'torchvision.transforms', 'torchvision.utils', 'lmdb', 'tensorboard',
                        'torch.utils.tensorboard', 'moviepy', 'moviepy.editor', 'plotly', 'plotly.graph_objects',
                        'plotly.colors', 'ray', 'cv2')):
    """Packages of the reference's environment that are absent here become empty stand-in modules (any attribute is a
    do-nothing class), except for the three things the TRAINING script really calls through them, which get working
    stand-ins (gangealing_amd/_standins.py): the VGG16 feature stack, make_grid and a scalar-logging SummaryWriter.
    Installed packages are never touched."""
    from gangealing_amd import _standins
    for name in names:
        if name in sys.modules:
            continue
        try:
            importlib.import_module(name)
        except Exception:
            m = _Stub(name)
            m.__path__ = []
            sys.modules[name] = m
            parent, _, leaf = name.rpartition('.')
            if isinstance(sys.modules.get(parent), _Stub):        # `from torchvision import models` must find the stub
                setattr(sys.modules[parent], leaf, m)
            if name == 'torchvision.models':
                m.vgg16 = _standins.vgg16
            elif name == 'torchvision.utils':
                m.make_grid, m.save_image = _standins.make_grid, _standins.save_image
            elif name == 'torch.utils.tensorboard':
                m.SummaryWriter = _standins.SummaryWriter


def _reference_module(root, target):
    """The reference's OWN module `target` (e.g. models.losses.lpips), loaded from its file under a private name - what a
    `--modules` shim falls back to for names this package does not provide."""
    import importlib.util
    key = '_gangealing_reference.' + target
    if key not in sys.modules:
        path = os.path.join(root, *target.split('.')) + '.py'
        spec = importlib.util.spec_from_file_location(key, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[key] = mod
        spec.loader.exec_module(mod)
    return sys.modules[key]


def inject(reference_root, modules=False):
    """Make `reference_root` importable with the HIP operator modules standing in for its own; modules=True: also this
    package's generator / STN / loss modules for the reference's (see the module docstring)."""
    reference_root = os.path.abspath(reference_root)
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
    stub_missing()
    # parent packages must exist as real packages of the reference so that sibling modules import normally
    for target, ours in OP_MODULES.items():
        sys.modules[target] = importlib.import_module(ours)
    done = sorted(OP_MODULES)
    if modules:
        for target, ours in MODEL_MODULES.items():
            if isinstance(ours, str):
                sys.modules[target] = importlib.import_module(ours)
            else:                                  # one module of ours holds what the reference keeps in two files
                src = importlib.import_module(ours[0])
                shim = types.ModuleType(target, f'{target}: names of {ours[0]} (gangealing_amd.launch --modules)')
                for name in ours[1]:
                    setattr(shim, name, getattr(src, name))
                # everything else the reference's file exposes (normalize_tensor, spatial_average, upsample, the
                # backbones module `pn`, ...) resolves to the reference's own code, loaded on first use
                # (dunder probes - __path__, __file__, __wrapped__ from the import machinery / inspect's hasattr -
                # must not execute the reference's file: an import error there would leave a hasattr() as something
                # other than AttributeError)
                def _fallback(name, _t=target):
                    if name.startswith('__'):
                        raise AttributeError(name)
                    return getattr(_reference_module(reference_root, _t), name)
                shim.__getattr__ = _fallback
                sys.modules[target] = shim
        done += sorted(MODEL_MODULES)
    return done


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    modules = os.environ.get('GANGEALING_LAUNCH_MODULES', '0') not in ('', '0')
    if argv and argv[0] == '--modules':
        modules, argv = True, argv[1:]
    if not argv:
        raise SystemExit(__doc__)
    script = os.path.abspath(argv[0])
    inject(os.path.dirname(script), modules=modules)
    sys.argv = [script] + list(argv[1:])
    runpy.run_path(script, run_name='__main__')


if __name__ == '__main__':
    main()
