// LPIPS / SimCLR-VGG perceptual distance tail, one feature tap per call
// (reference models/losses/lpips.py:26-28 normalize_tensor, :190-199 diffs / lins / spatial_average):
//   u = f / (sqrt(sum_c f^2) + eps)                     per pixel, both images
//   out[n] = 1/(H*W) * sum_pixels sum_c lin[c] * (u0[c] - u1[c])^2      (lin = NULL: all ones = lpips=False branch)
// feats (2N, C, H, W): samples [0,N) are image 0, [N,2N) image 1 (one batched backbone pass).
// As torch ops this is ~10 element-wise / reduction passes over the feature tensors forward and ~20 backward;
// here the forward reads the features twice (norms, then the weighted squared difference - the second read hits the
// Infinity Cache) and the backward reads them twice (moments, then the gradient) and writes the gradient once.  HBM-bound; lanes run along the
// pixel axis (contiguous in NCHW), channels are a strided loop.
#include "../../include/gangealing_hip.h"
#include "gg_common.h"

namespace {

// Block = 64 lanes x CGN waves.  A lane owns VEC consecutive pixels (contiguous in NCHW: VEC = 4 -> 16-byte loads on the
// large taps), wave g the channels g, g + CGN, ...; per-pixel sums are combined through LDS in group order.  The channel
// loop is unrolled by U with all its loads issued first: round 3's form (one 4-byte load per lane and iteration, one
// iteration in flight) ran the 64-channel 128^2 tap at 2.5 TB/s and was latency-bound on the 512-channel 16^2 / 8^2 taps
// (64 / 16 blocks walking 128 channels each: 54 - 136 us for 4 - 17 MB).  CGN = 16 (1024 threads) on the small taps
// spreads the channel walk over four times as many lanes.
constexpr int PIX = 64, U = 4;

template <int NV, int CGN, int W>
__device__ __forceinline__ void pixel_sums(float (&v)[NV][W], float (*red)[NV][PIX * W], int px, int grp) {
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < W; ++e) red[grp][i][px * W + e] = v[i][e];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < W; ++e) {
      float t = red[0][i][px * W + e];
#pragma unroll
      for (int q = 1; q < CGN; ++q) t += red[q][i][px * W + e];
      v[i][e] = t;
    }
  __syncthreads();
}

template <int W>
struct PixVec;
template <>
struct PixVec<1> {
  static __device__ __forceinline__ void load(float (&d)[1], const float* p) { d[0] = *p; }
  static __device__ __forceinline__ void store(float* p, const float (&d)[1]) { *p = d[0]; }
};
template <>
struct PixVec<4> {
  static __device__ __forceinline__ void load(float (&d)[4], const float* p) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&d)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(d[0], d[1], d[2], d[3]);
  }
};

// sum over the block (CGN waves) in wave order; valid in thread 0.  smem: >= CGN floats
template <int CGN>
__device__ __forceinline__ float block_sum_waves(float v, float* smem) {
  v = gg::wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < CGN; ++q) r += smem[q];
  }
  __syncthreads();
  return r;
}

// gg::ordered_grid_sum for blocks of 64 * CGN threads (same hand-off protocol, same fixed summation order)
template <int CGN>
__device__ __forceinline__ bool ordered_sample_sum(float& v, float* part, unsigned* ticket, int row, int blk, int nblk,
                                                   float* smem) {
  if (nblk == 1) return threadIdx.x == 0;
  __shared__ unsigned last_arriver;
  if (threadIdx.x == 0) {
    __hip_atomic_store(part + (size_t)row * nblk + blk, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned t = __hip_atomic_fetch_add(ticket + row, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_arriver = (t == (unsigned)(nblk - 1)) ? 1u : 0u;
    if (t == (unsigned)(nblk - 1)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (!last_arriver) return false;
  float acc = 0.f;
  for (int b = threadIdx.x; b < nblk; b += 64 * CGN)
    acc += __hip_atomic_load(part + (size_t)row * nblk + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v = block_sum_waves<CGN>(acc, smem);
  if (threadIdx.x == 0) __hip_atomic_store(ticket + row, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return threadIdx.x == 0;
}

template <int VEC, int CGN>
__global__ __launch_bounds__(64 * CGN) void lpips_tail_fwd_kernel(float* __restrict__ out, const float* __restrict__ f,
                                                                  const float* __restrict__ lin, int n, int c,
                                                                  long long hw, float eps, float inv_hw, float* part,
                                                                  unsigned* ticket) {
  __shared__ float red[CGN][2][PIX * VEC];
  __shared__ float redw[CGN];
  const int s = blockIdx.y, px = threadIdx.x & (PIX - 1), grp = threadIdx.x >> 6;
  const float* f0 = f + (size_t)s * c * hw;
  const float* f1 = f + (size_t)(s + n) * c * hw;
  float acc = 0.f;
  for (long long p0 = (long long)blockIdx.x * (PIX * VEC); p0 < hw; p0 += (long long)gridDim.x * (PIX * VEC)) {
    const long long p = p0 + px * VEC;
    const bool ok = p < hw;                                  // hw % VEC == 0: a lane's pixels are in or out together
    float v[2][VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[0][e] = v[1][e] = 0.f;
    if (ok)
      for (int k = grp; k < c; k += CGN * U) {
        float a[U][VEC], b[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int kk = k + u * CGN;
          const size_t off = (size_t)(kk < c ? kk : k) * hw + p;
          PixVec<VEC>::load(a[u], f0 + off);
          PixVec<VEC>::load(b[u], f1 + off);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (k + u * CGN < c) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) { v[0][e] += a[u][e] * a[u][e]; v[1][e] += b[u][e] * b[u][e]; }
          }
      }
    pixel_sums<2, CGN, VEC>(v, red, px, grp);
    float a0[VEC], a1[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { a0[e] = 1.f / (sqrtf(v[0][e]) + eps); a1[e] = 1.f / (sqrtf(v[1][e]) + eps); }
    if (ok)
      for (int k = grp; k < c; k += CGN * U) {
        float a[U][VEC], b[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int kk = k + u * CGN;
          const size_t off = (size_t)(kk < c ? kk : k) * hw + p;
          PixVec<VEC>::load(a[u], f0 + off);
          PixVec<VEC>::load(b[u], f1 + off);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (k + u * CGN < c) {
            const float l = lin ? lin[k + u * CGN] : 1.f;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              const float t = a[u][e] * a0[e] - b[u][e] * a1[e];
              acc += l * t * t;
            }
          }
      }
  }
  // the sample's blocks are summed in a fixed order (no float atomics: reproducible)
  float tot = block_sum_waves<CGN>(acc, redw);
  if (ordered_sample_sum<CGN>(tot, part, ticket, s, blockIdx.x, gridDim.x, redw)) out[s] = tot * inv_hw;
}

// df0[c] = a0 * q[c] - u0[c] * (sum_k q[k] u0[k]) / n0,   q[c] = 2 g/(HW) lin[c] (u0[c] - u1[c]);  df1 with -q.
template <int VEC, int CGN>
__global__ __launch_bounds__(64 * CGN) void lpips_tail_bwd_kernel(float* __restrict__ df, const float* __restrict__ f,
                                                                  const float* __restrict__ lin,
                                                                  const float* __restrict__ gout, int n, int c,
                                                                  long long hw, float eps, float inv_hw, int accumulate) {
  __shared__ float red[CGN][5][PIX * VEC];
  const int s = blockIdx.y, px = threadIdx.x & (PIX - 1), grp = threadIdx.x >> 6;
  const float* f0 = f + (size_t)s * c * hw;
  const float* f1 = f + (size_t)(s + n) * c * hw;
  float* d0 = df + (size_t)s * c * hw;
  float* d1 = df + (size_t)(s + n) * c * hw;
  const float g2 = 2.f * gout[s] * inv_hw;
  for (long long p0 = (long long)blockIdx.x * (PIX * VEC); p0 < hw; p0 += (long long)gridDim.x * (PIX * VEC)) {
    const long long p = p0 + px * VEC;
    const bool ok = p < hw;
    // one pass gives the norms AND the two projections t0 = sum_k q[k] u0[k], t1 = -sum_k q[k] u1[k] through the
    // lin-weighted moments (t0 = g2 (a0^2 W00 - a0 a1 W01)); their cancellation error is second order in |u0 - u1|
    float v[5][VEC];                                         // s00, s11, w00, w11, w01
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[i][e] = 0.f;
    if (ok)
      for (int k = grp; k < c; k += CGN * U) {
        float a[U][VEC], b[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int kk = k + u * CGN;
          const size_t off = (size_t)(kk < c ? kk : k) * hw + p;
          PixVec<VEC>::load(a[u], f0 + off);
          PixVec<VEC>::load(b[u], f1 + off);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (k + u * CGN < c) {
            const float l = lin ? lin[k + u * CGN] : 1.f;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              const float aa = a[u][e], bb = b[u][e];
              v[0][e] += aa * aa;
              v[1][e] += bb * bb;
              v[2][e] += l * aa * aa;
              v[3][e] += l * bb * bb;
              v[4][e] += l * aa * bb;
            }
          }
      }
    pixel_sums<5, CGN, VEC>(v, red, px, grp);
    float a0[VEC], a1[VEC], r0[VEC], r1[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float n0 = sqrtf(v[0][e]), n1 = sqrtf(v[1][e]);
      a0[e] = 1.f / (n0 + eps);
      a1[e] = 1.f / (n1 + eps);
      const float t0 = g2 * (a0[e] * a0[e] * v[2][e] - a0[e] * a1[e] * v[4][e]);
      const float t1 = g2 * (a1[e] * a1[e] * v[3][e] - a0[e] * a1[e] * v[4][e]);
      r0[e] = n0 > 0.f ? t0 / n0 : 0.f;
      r1[e] = n1 > 0.f ? t1 / n1 : 0.f;
    }
    if (ok)
      for (int k = grp; k < c; k += CGN * U) {
        float a[U][VEC], b[U][VEC], o0[U][VEC], o1[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int kk = k + u * CGN;
          const size_t off = (size_t)(kk < c ? kk : k) * hw + p;
          PixVec<VEC>::load(a[u], f0 + off);
          PixVec<VEC>::load(b[u], f1 + off);
          if (accumulate) {      // df already holds the gradient that reached this feature map from the next stage
            PixVec<VEC>::load(o0[u], d0 + off);
            PixVec<VEC>::load(o1[u], d1 + off);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (k + u * CGN < c) {
            const float l = g2 * (lin ? lin[k + u * CGN] : 1.f);
            float e0[VEC], e1[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              const float u0 = a[u][e] * a0[e], u1 = b[u][e] * a1[e];
              const float q = l * (u0 - u1);
              e0[e] = a0[e] * q - u0 * r0[e];
              e1[e] = -a1[e] * q - u1 * r1[e];
              if (accumulate) { e0[e] += o0[u][e]; e1[e] += o1[u][e]; }
            }
            const size_t off = (size_t)(k + u * CGN) * hw + p;
            PixVec<VEC>::store(d0 + off, e0);
            PixVec<VEC>::store(d1 + off, e1);
          }
      }
  }
}

// ------------------------------------------------------------------------------------------------
// 2x2 / stride-2 max pooling of the VGG16 trunk (lpips_backbones.py:101-121 via torchvision's features; ATen
// max_pool2d semantics: the window is scanned row-major and a later element replaces the maximum only if it is
// strictly greater or NaN).  The forward also writes the winner's position (0..3) as one byte per output, so the
// backward is a pure scatter-free stream: dx (4 B x 4 per output) is written once from dy and the code - no int64
// index tensor (8 B per output), no zero fill, no atomics.  Even H and W; W % 4 == 0 takes the 16-byte path.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pool_window(float a, float b, float c, float d, float& m, unsigned char& code) {
  m = a; code = 0;
  if (b > m || b != b) { m = b; code = 1; }
  if (c > m || c != c) { m = c; code = 2; }
  if (d > m || d != d) { m = d; code = 3; }
}

template <bool VEC>
__global__ __launch_bounds__(256) void maxpool2x2_fwd_kernel(float* __restrict__ out, unsigned char* __restrict__ code,
                                                             const float* __restrict__ x, long long nunits, int oh,
                                                             int ow) {
  // unit = VEC ? two horizontally adjacent outputs : one output
  const int upr = VEC ? ow / 2 : ow;                       // units per output row
  const long long stride = (long long)gridDim.x * 256;
  for (long long u = (long long)blockIdx.x * 256 + threadIdx.x; u < nunits; u += stride) {
    const long long row = u / upr;                          // plane * oh + oy
    const int ux = (int)(u - row * upr);
    const float* r0 = x + (size_t)row * 2 * (2 * ow) + (VEC ? 4 * ux : 2 * ux);
    const float* r1 = r0 + 2 * ow;
    if (VEC) {
      const float4 a = *reinterpret_cast<const float4*>(r0), b = *reinterpret_cast<const float4*>(r1);
      float m0, m1;
      unsigned char c0, c1;
      pool_window(a.x, a.y, b.x, b.y, m0, c0);
      pool_window(a.z, a.w, b.z, b.w, m1, c1);
      *reinterpret_cast<float2*>(out + (size_t)row * ow + 2 * ux) = make_float2(m0, m1);
      *reinterpret_cast<uchar2*>(code + (size_t)row * ow + 2 * ux) = make_uchar2(c0, c1);
    } else {
      float m;
      unsigned char c;
      pool_window(r0[0], r0[1], r1[0], r1[1], m, c);
      out[(size_t)row * ow + ux] = m;
      code[(size_t)row * ow + ux] = c;
    }
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void maxpool2x2_bwd_kernel(float* __restrict__ dx, const float* __restrict__ dy,
                                                             const unsigned char* __restrict__ code, long long nunits,
                                                             int oh, int ow) {
  const int upr = VEC ? ow / 2 : ow;
  const long long stride = (long long)gridDim.x * 256;
  for (long long u = (long long)blockIdx.x * 256 + threadIdx.x; u < nunits; u += stride) {
    const long long row = u / upr;
    const int ux = (int)(u - row * upr);
    float* r0 = dx + (size_t)row * 2 * (2 * ow) + (VEC ? 4 * ux : 2 * ux);
    float* r1 = r0 + 2 * ow;
    if (VEC) {
      const float2 g = *reinterpret_cast<const float2*>(dy + (size_t)row * ow + 2 * ux);
      const uchar2 c = *reinterpret_cast<const uchar2*>(code + (size_t)row * ow + 2 * ux);
      *reinterpret_cast<float4*>(r0) = make_float4(c.x == 0 ? g.x : 0.f, c.x == 1 ? g.x : 0.f, c.y == 0 ? g.y : 0.f,
                                                   c.y == 1 ? g.y : 0.f);
      *reinterpret_cast<float4*>(r1) = make_float4(c.x == 2 ? g.x : 0.f, c.x == 3 ? g.x : 0.f, c.y == 2 ? g.y : 0.f,
                                                   c.y == 3 ? g.y : 0.f);
    } else {
      const float g = dy[(size_t)row * ow + ux];
      const unsigned char c = code[(size_t)row * ow + ux];
      r0[0] = c == 0 ? g : 0.f; r0[1] = c == 1 ? g : 0.f;
      r1[0] = c == 2 ? g : 0.f; r1[1] = c == 3 ? g : 0.f;
    }
  }
}

// tile choice: 16-byte lanes on the large taps, 16 channel groups on the small ones (see the kernels' header)
struct TailPlan {
  int vec, cgn;
  dim3 grid;
};
TailPlan tail_plan(int n, int c, long long hw, const void* a, const void* b) {
  TailPlan t;
  const bool aligned = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
  t.vec = (hw % 4 == 0 && hw >= 16384 && aligned) ? 4 : 1;
  t.cgn = (hw <= 1024 && c >= 64) ? 16 : 4;
  long long bx = (hw + PIX * t.vec - 1) / (PIX * t.vec);
  if (bx > 8192) bx = 8192;
  t.grid = dim3((unsigned)bx, (unsigned)n);
  return t;
}

}  // namespace

extern "C" int gg_lpips_tail_fwd_f32(float* out, const float* feats, const float* lin, int n, int c, long long hw,
                                     float eps, void* stream) {
  if (n <= 0) return 0;
  if (!out || !feats || c <= 0 || hw <= 0 || n > 65535) return gg::fail(-2, "lpips_tail_fwd: bad arguments");
  hipStream_t st = gg::as_stream(stream);
  const TailPlan t = tail_plan(n, c, hw, feats, feats);
  const dim3 grid = t.grid;
  float* part = nullptr;
  unsigned* ticket = nullptr;
  if (grid.x > 1) {
    if (n > gg::kTickets) return gg::fail(-2, "lpips_tail_fwd: batch too large");
    part = reinterpret_cast<float*>(gg::scratch(st, sizeof(float) * (size_t)n * grid.x));
    ticket = gg::tickets(st);
    if (!part || !ticket) return -3;
  }
  const float inv = 1.f / (float)hw;
  if (t.vec == 4) lpips_tail_fwd_kernel<4, 4><<<grid, 256, 0, st>>>(out, feats, lin, n, c, hw, eps, inv, part, ticket);
  else if (t.cgn == 16) lpips_tail_fwd_kernel<1, 16><<<grid, 1024, 0, st>>>(out, feats, lin, n, c, hw, eps, inv, part, ticket);
  else lpips_tail_fwd_kernel<1, 4><<<grid, 256, 0, st>>>(out, feats, lin, n, c, hw, eps, inv, part, ticket);
  return gg::launch_status("lpips_tail_fwd");
}

extern "C" int gg_lpips_tail_bwd_f32(float* dfeats, const float* feats, const float* lin, const float* grad_out, int n,
                                     int c, long long hw, float eps, int accumulate, void* stream) {
  if (n <= 0) return 0;
  if (!dfeats || !feats || !grad_out || c <= 0 || hw <= 0 || n > 65535)
    return gg::fail(-2, "lpips_tail_bwd: bad arguments");
  hipStream_t st = gg::as_stream(stream);
  const TailPlan t = tail_plan(n, c, hw, feats, dfeats);
  const float inv = 1.f / (float)hw;
  if (t.vec == 4)
    lpips_tail_bwd_kernel<4, 4><<<t.grid, 256, 0, st>>>(dfeats, feats, lin, grad_out, n, c, hw, eps, inv, accumulate);
  else if (t.cgn == 16)
    lpips_tail_bwd_kernel<1, 16><<<t.grid, 1024, 0, st>>>(dfeats, feats, lin, grad_out, n, c, hw, eps, inv, accumulate);
  else
    lpips_tail_bwd_kernel<1, 4><<<t.grid, 256, 0, st>>>(dfeats, feats, lin, grad_out, n, c, hw, eps, inv, accumulate);
  return gg::launch_status("lpips_tail_bwd");
}

extern "C" int gg_maxpool2x2_fwd_f32(float* out, unsigned char* code, const float* x, long long planes, int h, int w,
                                     void* stream) {
  if (planes <= 0) return 0;
  if (!out || !code || !x || h <= 0 || w <= 0 || (h & 1) || (w & 1)) return gg::fail(-2, "maxpool2x2: even H, W required");
  const int oh = h / 2, ow = w / 2;
  const bool vec = (w % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(code) & 1) == 0;
  const long long nunits = planes * oh * (vec ? ow / 2 : ow);
  hipStream_t st = gg::as_stream(stream);
  if (vec) maxpool2x2_fwd_kernel<true><<<gg::stream_grid(nunits, 256), 256, 0, st>>>(out, code, x, nunits, oh, ow);
  else maxpool2x2_fwd_kernel<false><<<gg::stream_grid(nunits, 256), 256, 0, st>>>(out, code, x, nunits, oh, ow);
  return gg::launch_status("maxpool2x2_fwd");
}

extern "C" int gg_maxpool2x2_bwd_f32(float* dx, const float* dy, const unsigned char* code, long long planes, int h,
                                     int w, void* stream) {
  if (planes <= 0) return 0;
  if (!dx || !dy || !code || h <= 0 || w <= 0 || (h & 1) || (w & 1)) return gg::fail(-2, "maxpool2x2: even H, W required");
  const int oh = h / 2, ow = w / 2;
  const bool vec = (w % 4 == 0) && ((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(code) & 1) == 0;
  const long long nunits = planes * oh * (vec ? ow / 2 : ow);
  hipStream_t st = gg::as_stream(stream);
  if (vec) maxpool2x2_bwd_kernel<true><<<gg::stream_grid(nunits, 256), 256, 0, st>>>(dx, dy, code, nunits, oh, ow);
  else maxpool2x2_bwd_kernel<false><<<gg::stream_grid(nunits, 256), 256, 0, st>>>(dx, dy, code, nunits, oh, ow);
  return gg::launch_status("maxpool2x2_bwd");
}
