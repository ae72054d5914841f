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

// Per-(device, stream) scratch memory owned by the library (gg_runtime.hip): partial results of split reductions.
// scratch(): >= bytes, 256-byte aligned, contents undefined; nullptr on failure (gg_last_error is set).
// tickets(): kTickets zero-initialised counters; every kernel that uses them leaves them zero again.
void* scratch(hipStream_t st, size_t bytes);
constexpr int kTickets = 16384;
unsigned* tickets(hipStream_t st);

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

// Deterministic finish of a sum that is spread over the `nblk` blocks of one "row" of the grid (a channel, a sample,
// the whole launch).  Block-collective for 256-thread blocks; v[] holds the block's NV totals in thread 0.  They are
// stored to part[(row * nblk + blk) * NV ..], and the block that arrives LAST at the row's ticket counter re-reads
// all of the row's partials (thread t takes blocks t, t + 256, ..., then the fixed tree of block_sum_256) - a fixed
// summation order, whatever order the blocks ran in (one atomic add per block would combine them in arrival order: a
// different rounding every run).  Returns true in thread 0 of that last block, with v[] = the totals; the counter is
// left at zero for the next launch.  `smem`: >= 4 values of block-shared scratch.
// Hand-off protocol of the CDNA4 guide (inter-workgroup communication): agent-scope payload stores, an agent-scope
// RELEASE fence, THEN the ticket; the last arriver issues one agent-scope ACQUIRE (per-CU L1 is never refreshed by
// other CUs' stores) and reads with agent-scope loads.
template <typename T, int NV>
__device__ __forceinline__ bool ordered_grid_sum(T (&v)[NV], T* part, unsigned* ticket, int row, int blk, int nblk,
                                                 T* smem) {
  if (nblk == 1) return threadIdx.x == 0;
  __shared__ unsigned last_arriver;
  if (threadIdx.x == 0) {
    T* mine = part + ((size_t)row * nblk + blk) * NV;
#pragma unroll
    for (int i = 0; i < NV; ++i) __hip_atomic_store(mine + i, v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // release at agent scope: the payload is visible to the agent before the ticket is (the memory model's statement of
    // "drain the stores, then arrive"; round 3 relied on s_waitcnt vmcnt(0) + write-through stores)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned t = __hip_atomic_fetch_add(ticket + row, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_arriver = (t == (unsigned)(nblk - 1)) ? 1u : 0u;
    if (t == (unsigned)(nblk - 1)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (!last_arriver) return false;
  T* all = part + (size_t)row * nblk * NV;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    T acc = T(0);
    for (int b = threadIdx.x; b < nblk; b += 256)
      acc += __hip_atomic_load(all + (size_t)b * NV + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v[i] = block_sum_256<T>(acc, smem);
  }
  if (threadIdx.x == 0) __hip_atomic_store(ticket + row, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return threadIdx.x == 0;
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
