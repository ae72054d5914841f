/* TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) - never part of the product.
 *
 * Forced-include prelude (hipcc -include) that lets the REFERENCE's own CUDA source
 * utils/splat2d_cuda/src/splat_gpu_impl.cu compile unmodified, from where it lies under /root/reference, into
 * oracle/_ref/libsplat_ref.so: the file uses five CUDA runtime identifiers and nothing else (no THC, no torch).
 * The resulting library is the reference kernel itself, used as the checker for gangealing_amd's splat2d
 * (oracle/make_golden_splat.py, tests/test_gpu_splat.py).  The product has no CUDA names anywhere. */
#include <hip/hip_runtime.h>
#define cudaStream_t hipStream_t
#define cudaError_t hipError_t
#define cudaSuccess hipSuccess
#define cudaGetLastError hipGetLastError
#define cudaGetErrorString hipGetErrorString
