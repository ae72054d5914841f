"""CPU suite: the C-ABI library loads, exports every symbol include/gangealing_hip.h declares, the
ctypes prototypes agree with the header, and the product ops refuse CPU tensors (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import REPO


def header_decls():
    hdr = open(os.path.join(REPO, 'include', 'gangealing_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    return re.findall(r'(?:int|const char\*)\s+(gg_\w+)\s*\(([^;]*?)\)\s*;', hdr)


@pytest.fixture(scope='module')
def lib():
    import __graft_entry__ as ge
    from gangealing_amd import _lib
    if not _lib.available():
        ge.build()
    return _lib.load()


def test_exports_every_declared_symbol(lib):
    from gangealing_amd import _lib
    names = [n for n, _ in header_decls()]
    assert len(names) >= 28
    for n in names:
        assert hasattr(lib, n), f'{n} declared in the header but not exported'
    assert set(_lib.exported_symbols()) == set(names)
    assert lib.gg_build_arch() == b'gfx950'
    assert lib.gg_abi_version() == _lib.ABI_VERSION


def test_ctypes_prototypes_match_header():
    from gangealing_amd import _lib
    tm = {'float*': 'p', 'const float*': 'p', 'double*': 'p', 'const double*': 'p', 'unsigned short*': 'p', 'unsigned char*': 'p', 'const unsigned char*': 'p', 'const signed char*': 'p',
          'const unsigned short*': 'p', 'const void*': 'p', 'unsigned int*': 'p', 'const unsigned int*': 'p', 'int*': 'p', 'const int*': 'p', 'int': 'i', 'long long': 'q',
          'float': 'f', 'double': 'd', 'void*': 's'}
    for name, args in header_decls():
        if name in ('gg_abi_version', 'gg_last_error', 'gg_build_arch', 'gg_scratch_release', 'gg_set_allocator', 'gg_last_conv_kernel', 'gg_set_tuning', 'gg_last_sign_bits_written', 'gg_last_amax_written',
                    'gg_blur4_bits_words'):
            continue
        proto = ''.join(tm[re.sub(r'\s+\w+$', '', a.strip()).replace(' *', '*')] for a in args.split(','))
        assert _lib._PROTOS[name] == proto, name


def test_reference_splat_symbol_is_exported(lib):
    """The reference's one true C symbol (utils/splat2d_cuda/src/splat_gpu_impl.cuh:11-22): same name, stream first,
    void return - so splat_gpu.c:29-31 links unchanged.  Declared in the header with exactly those eleven parameters."""
    assert hasattr(lib, 'SplatForwardGpu')
    hdr = open(os.path.join(REPO, 'include', 'gangealing_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    m = re.search(r'void\s+SplatForwardGpu\s*\(([^;]*?)\)\s*;', hdr)
    assert m, 'SplatForwardGpu not declared'
    args = [re.sub(r'\s+\w+$', '', a.strip()).replace(' *', '*') for a in m.group(1).split(',')]
    assert args == ['void*', 'const float*', 'const float*', 'const float*', 'float*', 'float*',
                    'const int', 'const int', 'const int', 'const int', 'const int']
    # top_count <= 0: nothing to do, nothing launched, no GPU needed
    lib.SplatForwardGpu.restype = None
    lib.SplatForwardGpu.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 5
    lib.SplatForwardGpu(None, None, None, None, None, None, 0, 0, 0, 0, 0)


def test_argument_errors_do_not_need_a_gpu(lib):
    # negative sizes are rejected before any launch
    rc = lib.gg_upfirdn2d_f32(None, None, None, 1, 4, 4, 4, 4, 0, 1, 1, 1, 0, 0, 0, 0, None)
    assert rc != 0 and b'upfirdn2d' in lib.gg_last_error()
    rc = lib.gg_conv2d_f32(ctypes.c_void_p(8), ctypes.c_void_p(8), ctypes.c_void_p(8), None, None, None,
                           1, 1, 4, 4, 8, 8, 5, 1, 0, 0, 0, 0, None)
    assert rc != 0 and b'kernel size' in lib.gg_last_error()


def test_no_cpu_fallback():
    from gangealing_amd import _lib
    from gangealing_amd.op import upfirdn2d, fused_leaky_relu, conv2d_gradfix
    from gangealing_amd.spatial_transformers.antialiased_sampling import MipmapWarp, BilinearDownsample
    x = torch.zeros(1, 2, 8, 8)
    with pytest.raises(_lib.HipLibraryError):
        upfirdn2d(x, torch.ones(4, 4))
    with pytest.raises(_lib.HipLibraryError):
        fused_leaky_relu(x, torch.zeros(2))
    with pytest.raises(_lib.HipLibraryError):
        conv2d_gradfix.conv2d(x, torch.zeros(2, 2, 3, 3), padding=1)
    with pytest.raises(_lib.HipLibraryError):
        MipmapWarp(3.5)(x, torch.zeros(1, 4, 4, 2))
    with pytest.raises(_lib.HipLibraryError):
        BilinearDownsample(2, 2)(x)


def test_product_never_imports_oracle():
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, 'gangealing_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(root, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M):
                    bad.append(f)
    assert not bad, bad
