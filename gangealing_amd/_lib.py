"""ctypes binding of libgangealing_hip.so (C ABI declared in include/gangealing_hip.h).

The product path is HIP-only: if the library is missing or a tensor is not on a HIP device the
call raises - there is no CPU fallback (the CPU restatement lives in oracle/ and is test-only).
PyTorch is used purely as plumbing: device memory (tensor.data_ptr()), the current HIP stream
(torch.cuda.current_stream().cuda_stream) and torch.distributed (RCCL).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# GANGEALING_HIP_LIB selects another build of the same library (kernel A/B measurements); the ABI check still applies.
LIB_PATH = os.environ.get('GANGEALING_HIP_LIB') or os.path.join(_HERE, 'lib', 'libgangealing_hip.so')
ABI_VERSION = 6
NOT_SERVED = -1000            # GG_NOT_SERVED of the header

# signature alphabet: p device pointer (tensor or None), i int, q long long, f float, d double, s stream
_PROTOS = {
    'gg_fused_bias_act_f32': 'ppppiiffqqis',
    'gg_fused_bias_act_f64': 'ppppiiddqqis',
    'gg_fused_bias_act_f16': 'ppppiiffqqis',
    'gg_fused_lrelu_bwd_f32': 'ppppffiiqs',
    'gg_fused_lrelu_bwd_f64': 'ppppddiiqs',
    'gg_fused_lrelu_bwd_acc_f32': 'ppppffiiqis',
    'gg_fused_lrelu_bwd_f16': 'ppppffiiqs',
    'gg_noise_bias_act_f32': 'pppppffiiqs',
    'gg_upfirdn2d_f32': 'pppiiiiiiiiiiiiis',
    'gg_upfirdn2d_f64': 'pppiiiiiiiiiiiiis',
    'gg_upfirdn2d_f16': 'pppiiiiiiiiiiiiis',
    'gg_blur4_fused_f32': 'pppiiiiiiiippppffs',
    'gg_blur4_fused_bits_f32': 'pppiiiiiiiippppffs',
    'gg_blur4_act_bwd_f32': 'pppiiiiiiiipffpis',
    'gg_splat_forward_f32': 'pppppiiiiis',
    'gg_splat2d_f32': 'ppppppiiiiiis',
    'gg_mip_downsample2x_f32': 'ppiiis',
    'gg_mip_downsample2x_bwd_f32': 'ppiiis',
    'gg_mipmap_warp_fwd_f32': 'ppppipiiiiiiiiiffiis',
    'gg_mipmap_warp_bwd_f32': 'ppppppipiiiiiiiiiffiips',
    'gg_mipmap_warp_indices_f32': 'ppppppiiiiiffiis',
    'gg_similarity_matrix_f32': 'ppiis',
    'gg_similarity_matrix_bwd_f32': 'pppiis',
    'gg_affine_grid_f32': 'ppiiis',
    'gg_affine_grid_bwd_f32': 'ppiiis',
    'gg_flow_compose_fwd_f32': 'pppppiiiis',
    'gg_flow_compose_bwd_f32': 'ppppppppiiiis',
    'gg_flow_resize_f32': 'ppiiiiifs',
    'gg_flow_resize_bwd_f32': 'ppiiiiifs',
    'gg_bilinear_downsample_f32': 'ppiiiis',
    'gg_bilinear_downsample_bwd_f32': 'ppiiiis',
    'gg_flow_losses_f32': 'ppiiis',
    'gg_flow_losses_bwd_f32': 'pppiiis',
    'gg_conv_pack_weight_f32': 'ppiiiiiiifs',
    'gg_conv2d_f32': 'ppppppiiiiiiiiiiiis',
    'gg_conv_pack_weight_split': 'ppiiiiiiifis',
    'gg_conv3x3_masked_dgrad_f32': 'pppffpqippiiiiis',
    'gg_conv_pack_weights_many': 'pis',
    'gg_conv2d_split_f32': 'pppqipppiiiiiiiiiiiis',
    'gg_conv2d_split_act_f32': 'pppqipppffiiiiiiiiiiiis',
    'gg_conv1x1_split_residual_f32': 'pppqippppiiiiiiis',
    'gg_modconv3x3_act_f32': 'ppppqipppppffiiiiis',
    'gg_modconv3x3_act_bits_f32': 'ppppqipppppffiiiiips',
    'gg_conv3x3_masked_dgrad_bits_f32': 'pppffpqippiiiiis',
    'gg_modconv3x3_act_amax_f32': 'ppppqipppppffiiiiipps',
    'gg_conv3x3_fewout_masked_bits_f32': 'pppffpiiiiis',
    'gg_torgb_limb_f32': 'pppppppppiiqs',
    'gg_convT3x3s2_prelimb_f32': 'ppppqppiiiiiiiis',
    'gg_conv2d_wgrad_f32': 'pppiiiiiiiiifs',
    'gg_conv2d_wgrad_split_f32': 'pppiiiiiiiiifis',
    'gg_conv2d_wgrad_acc_f32': 'pppiiiiiiiiifis',
    'gg_conv2d_wgrad_ws_f32': 'pppiiiiiiiiifiipqs',
    'gg_conv3x3_masked_wgrad_f32': 'pppppffiiiiifiipqs',
    'gg_style_demod_f32': 'pppqpppiiiifffs',
    'gg_lpips_tail_fwd_f32': 'pppiiqfs',
    'gg_lpips_tail_bwd_f32': 'ppppiiqfis',
    'gg_maxpool2x2_fwd_f32': 'pppqiis',
    'gg_maxpool2x2_bwd_f32': 'pppqiis',
    'gg_torgb_dgrad_add_f32': 'ppppfiiqs',
    'gg_plane_dot_f32': 'pppiqs',
    'gg_modconv_style_grad_f32': 'ppppppiiis',
    'gg_adam_ema_f32': 'pppppqffffiffs',
    'gg_adam_ema_dev_f32': 'pppppqpffffs',
    'gg_upfirdn2d_add_f32': 'ppppiiiiiiiiiiiiis',
    'gg_add_scale_f32': 'pppfqs',
    'gg_style_bank_f32': 'ppqipiipiiis',
    'gg_scratch_reserve': 'qs',
}
_CTYPE = {'p': ctypes.c_void_p, 'i': ctypes.c_int, 'q': ctypes.c_longlong, 'f': ctypes.c_float,
          'd': ctypes.c_double, 's': ctypes.c_void_p}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def exported_symbols():
    return ['gg_abi_version', 'gg_last_error', 'gg_build_arch', 'gg_scratch_release', 'gg_set_allocator',
            'gg_last_conv_kernel', 'gg_set_tuning', 'gg_last_sign_bits_written', 'gg_last_amax_written',
            'gg_blur4_bits_words'] + sorted(_PROTOS)


def load():
    """Load the shared library (once).  Raises HipLibraryError if it is missing or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'(hipcc --offload-arch=gfx950).  gangealing_amd has no CPU fallback.')
    lib = ctypes.CDLL(LIB_PATH)
    lib.gg_abi_version.restype = ctypes.c_int
    lib.gg_last_error.restype = ctypes.c_char_p
    lib.gg_build_arch.restype = ctypes.c_char_p
    lib.gg_scratch_release.restype = ctypes.c_int
    if hasattr(lib, 'gg_last_conv_kernel'):          # (absent from A/B builds of older kernel sources)
        lib.gg_last_conv_kernel.restype = ctypes.c_char_p
        lib.gg_last_conv_kernel.argtypes = []
    lib.gg_scratch_release.argtypes = []
    if lib.gg_abi_version() != ABI_VERSION:
        raise HipLibraryError(f'ABI mismatch: library {lib.gg_abi_version()} != python {ABI_VERSION}; rebuild')
    for name, proto in _PROTOS.items():
        fn = getattr(lib, name)
        fn.restype = ctypes.c_int
        fn.argtypes = [_CTYPE[c] for c in proto]
    _install_allocator(lib)
    _lib = lib
    return lib


# The library's scratch (partial results of split reductions) comes out of torch's caching allocator: no hipMalloc
# outside torch's accounting (which fails once torch has reserved most of HBM), and growth while a stream is being
# captured is legal (torch serves it from the graph's private pool; the stream's ticket page is the exception: the
# library refuses to create it inside a capture - `reserve_scratch(0)` or one eager step first).  The tensors are kept alive here until the library
# hands the pointer back (gg_scratch_release); allocation happens on torch's current stream = the stream `call` launches on.
_ALLOC_FN = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_longlong)
_FREE_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p)
_SCRATCH_TENSORS = {}
_CALLBACKS = []


def _install_allocator(lib):
    if os.environ.get('GANGEALING_SCRATCH_HIPMALLOC') == '1' or not torch.cuda.is_available():
        return

    def alloc(nbytes):
        try:
            t = torch.empty(int(nbytes), dtype=torch.uint8, device=torch.device('cuda', torch.cuda.current_device()))
        except Exception:            # noqa: BLE001 - reported by the library as an allocation failure
            return None
        _SCRATCH_TENSORS[t.data_ptr()] = t
        return t.data_ptr()

    def free(ptr):
        _SCRATCH_TENSORS.pop(ptr, None)

    cbs = (_ALLOC_FN(alloc), _FREE_FN(free))
    _CALLBACKS.append(cbs)           # ctypes callbacks must outlive the library's use of them
    lib.gg_set_allocator.restype = ctypes.c_int
    lib.gg_set_allocator.argtypes = [_ALLOC_FN, _FREE_FN]
    rc = lib.gg_set_allocator(*cbs)
    if rc != 0:
        raise HipLibraryError(f'gg_set_allocator failed: {lib.gg_last_error().decode()}')


class Strided:
    """Marks a tensor whose strides are passed to the entry point explicitly (skips the contiguity check)."""

    def __init__(self, tensor):
        self.tensor = tensor


def _dev_ptr(t):
    """-> the tensor's device address as a plain int (ctypes converts it for a c_void_p parameter), or None."""
    if t is None:
        return None
    if isinstance(t, Strided):
        if not t.tensor.is_cuda:
            raise HipLibraryError('gangealing_amd operators run on HIP devices only')
        return t.tensor.data_ptr()
    if not isinstance(t, torch.Tensor):
        raise TypeError(f'expected a tensor or None, got {type(t)}')
    if not t.is_cuda:
        raise HipLibraryError('gangealing_amd operators run on HIP devices only (got a %s tensor); '
                              'the CPU restatement is test infrastructure under oracle/' % t.device.type)
    if not t.is_contiguous():
        raise HipLibraryError('internal error: non-contiguous tensor passed to the C ABI')
    return t.data_ptr()


_ENTRY = {}          # name -> (ctypes function, prototype string, argument count): resolved once per entry point
CALLS = 0            # entry-point calls so far (bench.py reports calls per step)
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _current_stream_handle():
    # the raw handle of torch's current stream without building a torch.cuda.Stream object per launch (the step issues
    # ~450 library calls; at per-GPU batch 5 the eager step is bound by the host)
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def call(name, *args, allow=()):
    """Invoke a C-ABI entry point on torch's current HIP stream; the trailing stream argument is
    supplied here.  Tensors are passed as raw device pointers.  Returns the status code; codes other than 0 raise
    unless listed in `allow` (e.g. NOT_SERVED = "shape not served, nothing launched" of the optional fused entry points)."""
    global CALLS
    CALLS += 1
    ent = _ENTRY.get(name)
    if ent is None:
        lib = load()
        proto = _PROTOS[name]
        ent = _ENTRY[name] = (getattr(lib, name), proto, len(proto) - 1)
    fn, proto, n = ent
    if len(args) != n:
        raise TypeError(f'{name}: expected {n} arguments, got {len(args)}')
    conv = [_dev_ptr(a) if c == 'p' else (int(a) if (c == 'i' or c == 'q') else float(a)) for a, c in zip(args, proto)]
    conv.append(_current_stream_handle())
    rc = fn(*conv)
    if rc != 0 and rc not in allow:
        raise HipLibraryError(f'{name} failed (code {rc}): {load().gg_last_error().decode()}')
    return rc


def available():
    return os.path.exists(LIB_PATH)


def reserve_scratch(nbytes=0):
    """Create the current stream's scratch buffer (>= nbytes) and its ticket page now.  The library refuses to create a
    ticket page inside a hipGraph capture (its clearing memset would only be recorded): call this - or run the step once
    eagerly - on every stream a capture will launch library kernels on."""
    return call('gg_scratch_reserve', int(nbytes))


TUNING_RESET = -2147483648      # GG_TUNING_RESET: back to the environment's / built-in setting


def set_tuning(name, value):
    """Measurement aid (gg_set_tuning): flip a kernel-selection switch of the convolution dispatcher at run time."""
    lib = load()
    lib.gg_set_tuning.argtypes = [ctypes.c_char_p, ctypes.c_int]
    rc = lib.gg_set_tuning(name.encode(), int(value))
    if rc != 0:
        raise HipLibraryError(f'gg_set_tuning({name}) failed: {lib.gg_last_error().decode()}')
