// Shared host/device helpers for the gfx950 (MI355X, CDNA4) GANgealing kernels.
// Wave = 64 lanes; 256 CUs in 8 XCDs (block b is observed to land on XCD b % 8).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gg {

// Error plumbing behind the C ABI: every entry point returns 0 or a non-zero code and leaves a
// message retrievable through gg_last_error().
int fail(int code, const char* fmt, ...);
int launch_status(const char* what);

constexpr int kWave = 64;
constexpr int kNumXcd = 8;
constexpr int kNumCu = 256;

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline unsigned ceil_div_u(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

// Grid for HBM-streaming kernels: enough blocks to fill 256 CUs x 8 blocks, grid-stride the rest.
inline unsigned stream_grid(long long work_items, int block) {
  long long b = (work_items + block - 1) / block;
  if (b < 1) b = 1;
  if (b > (long long)kNumCu * 8) b = (long long)kNumCu * 8;
  return (unsigned)b;
}

// Bijective XCD-aware remap: consecutive *logical* tiles are placed on the same XCD so that tiles
// sharing halo rows / operand panels hit the same 4 MiB L2.  Speed only - never correctness.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks) {
  const unsigned q = nblocks / kNumXcd, r = nblocks % kNumXcd;
  const unsigned xcd = bid % kNumXcd, idx = bid / kNumXcd;
  const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

__device__ __forceinline__ int floor_div(int a, int b) {   // b > 0; rounds toward -inf
  int c = a / b;
  return (c * b > a) ? c - 1 : c;
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;                                         // valid in lane 0
}

// Sum over a 256-thread block (4 waves); result valid in thread 0.  `smem` holds >= 4 values.
template <typename T>
__device__ __forceinline__ T block_sum_256(T v, T* smem) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  T r = 0;
  if (threadIdx.x == 0) r = smem[0] + smem[1] + smem[2] + smem[3];
  __syncthreads();
  return r;
}

// Raw buffer access: address = resource base + voffset (per lane) + soffset (scalar).  A voffset at or beyond the
// resource's num_records makes a load return 0 and a store vanish - kOobOffset selects that for masked lanes
// (resources are < 2 GiB), which keeps bounds handling out of the control flow.  The descriptor words go through
// readfirstlane so that the compiler can prove they are wave-uniform (otherwise every access is wrapped in a
// waterfall loop).
constexpr unsigned kOobOffset = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* p, int bytes) {
  const unsigned long long u = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                           __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ float buffer_load_f32(__amdgpu_buffer_rsrc_t r, unsigned voffset, int soffset) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voffset, soffset, 0));
}
__device__ __forceinline__ void buffer_store_f32(float v, __amdgpu_buffer_rsrc_t r, unsigned voffset, int soffset) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, (int)voffset, soffset, 0);
}

// Exactly-rounded fp32 ops that the compiler may not contract into FMAs: the sampling-index math
// must reproduce the IEEE sequence of the reference formulas bit for bit.
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub_rn(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float div_rn(float a, float b) { return __fdiv_rn(a, b); }

}  // namespace gg
