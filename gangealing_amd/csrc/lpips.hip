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

// Block = 64 pixels (lanes, contiguous in NCHW) x 4 channel groups (waves): wave g handles channels g, g+4, ...;
// per-pixel sums are combined through LDS.  (One thread per pixel walking all channels left the 8x8 / 16x16 taps with
// 16 blocks of serial 512-channel loops.)
constexpr int PIX = 64, CG = 4;

template <int NV>
__device__ __forceinline__ void pixel_sums(float (&v)[NV], float (*red)[NV][PIX], int px, int grp) {
#pragma unroll
  for (int i = 0; i < NV; ++i) red[grp][i][px] = v[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = red[0][i][px] + red[1][i][px] + red[2][i][px] + red[3][i][px];
  __syncthreads();
}

__global__ __launch_bounds__(256) void lpips_tail_fwd_kernel(float* __restrict__ out, const float* __restrict__ f,
                                                             const float* __restrict__ lin, int n, int c,
                                                             long long hw, float eps, float inv_hw, float* part,
                                                             unsigned* ticket) {
  __shared__ float red[CG][2][PIX];
  __shared__ float red4[4];
  const int s = blockIdx.y, px = threadIdx.x & (PIX - 1), grp = threadIdx.x >> 6;
  const float* f0 = f + (size_t)s * c * hw;
  const float* f1 = f + (size_t)(s + n) * c * hw;
  float acc = 0.f;
  for (long long p0 = (long long)blockIdx.x * PIX; p0 < hw; p0 += (long long)gridDim.x * PIX) {
    const long long p = p0 + px;
    const bool ok = p < hw;
    float v[2] = {0.f, 0.f};
    if (ok)
      for (int k = grp; k < c; k += CG) {
        const float a = f0[(size_t)k * hw + p], b = f1[(size_t)k * hw + p];
        v[0] += a * a;
        v[1] += b * b;
      }
    pixel_sums<2>(v, red, px, grp);
    const float a0 = 1.f / (sqrtf(v[0]) + eps), a1 = 1.f / (sqrtf(v[1]) + eps);
    if (ok)
      for (int k = grp; k < c; k += CG) {
        const float t = f0[(size_t)k * hw + p] * a0 - f1[(size_t)k * hw + p] * a1;
        acc += (lin ? lin[k] : 1.f) * t * t;
      }
  }
  // the sample's blocks are summed in a fixed order (no float atomics: reproducible)
  float v[1] = {gg::block_sum_256<float>(acc, red4)};
  if (gg::ordered_grid_sum<float, 1>(v, part, ticket, s, blockIdx.x, gridDim.x, red4)) out[s] = v[0] * inv_hw;
}

// df0[c] = a0 * q[c] - u0[c] * (sum_k q[k] u0[k]) / n0,   q[c] = 2 g/(HW) lin[c] (u0[c] - u1[c]);  df1 with -q.
__global__ __launch_bounds__(256) void lpips_tail_bwd_kernel(float* __restrict__ df, const float* __restrict__ f,
                                                             const float* __restrict__ lin,
                                                             const float* __restrict__ gout, int n, int c,
                                                             long long hw, float eps, float inv_hw, int accumulate) {
  __shared__ float red[CG][5][PIX];
  const int s = blockIdx.y, px = threadIdx.x & (PIX - 1), grp = threadIdx.x >> 6;
  const float* f0 = f + (size_t)s * c * hw;
  const float* f1 = f + (size_t)(s + n) * c * hw;
  float* d0 = df + (size_t)s * c * hw;
  float* d1 = df + (size_t)(s + n) * c * hw;
  const float g2 = 2.f * gout[s] * inv_hw;
  for (long long p0 = (long long)blockIdx.x * PIX; p0 < hw; p0 += (long long)gridDim.x * PIX) {
    const long long p = p0 + px;
    const bool ok = p < hw;
    // one pass gives the norms AND the two projections t0 = sum_k q[k] u0[k], t1 = -sum_k q[k] u1[k] through the
    // lin-weighted moments (t0 = g2 (a0^2 W00 - a0 a1 W01)); their cancellation error is second order in |u0 - u1|
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};                  // s00, s11, w00, w11, w01
    if (ok)
      for (int k = grp; k < c; k += CG) {
        const float a = f0[(size_t)k * hw + p], b = f1[(size_t)k * hw + p];
        const float l = lin ? lin[k] : 1.f;
        v[0] += a * a;
        v[1] += b * b;
        v[2] += l * a * a;
        v[3] += l * b * b;
        v[4] += l * a * b;
      }
    pixel_sums<5>(v, red, px, grp);
    const float n0 = sqrtf(v[0]), n1 = sqrtf(v[1]);
    const float a0 = 1.f / (n0 + eps), a1 = 1.f / (n1 + eps);
    const float t0 = g2 * (a0 * a0 * v[2] - a0 * a1 * v[4]);
    const float t1 = g2 * (a1 * a1 * v[3] - a0 * a1 * v[4]);
    const float r0 = n0 > 0.f ? t0 / n0 : 0.f, r1 = n1 > 0.f ? t1 / n1 : 0.f;
    if (ok)
      for (int k = grp; k < c; k += CG) {
        const float u0 = f0[(size_t)k * hw + p] * a0, u1 = f1[(size_t)k * hw + p] * a1;
        const float q = g2 * (lin ? lin[k] : 1.f) * (u0 - u1);
        const float e0 = a0 * q - u0 * r0, e1 = -a1 * q - u1 * r1;
        if (accumulate) {        // df already holds the gradient that reached this feature map from the next stage
          d0[(size_t)k * hw + p] += e0;
          d1[(size_t)k * hw + p] += e1;
        } else {
          d0[(size_t)k * hw + p] = e0;
          d1[(size_t)k * hw + p] = e1;
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

dim3 tail_grid(int n, long long hw) {
  long long bx = (hw + PIX - 1) / PIX;
  if (bx > 8192) bx = 8192;
  return dim3((unsigned)bx, (unsigned)n);
}

}  // namespace

extern "C" int gg_lpips_tail_fwd_f32(float* out, const float* feats, const float* lin, int n, int c, long long hw,
                                     float eps, void* stream) {
  if (n <= 0) return 0;
  if (!out || !feats || c <= 0 || hw <= 0 || n > 65535) return gg::fail(-2, "lpips_tail_fwd: bad arguments");
  hipStream_t st = gg::as_stream(stream);
  const dim3 grid = tail_grid(n, hw);
  float* part = nullptr;
  unsigned* ticket = nullptr;
  if (grid.x > 1) {
    if (n > gg::kTickets) return gg::fail(-2, "lpips_tail_fwd: batch too large");
    part = reinterpret_cast<float*>(gg::scratch(st, sizeof(float) * (size_t)n * grid.x));
    ticket = gg::tickets(st);
    if (!part || !ticket) return -3;
  }
  lpips_tail_fwd_kernel<<<grid, 256, 0, st>>>(out, feats, lin, n, c, hw, eps, 1.f / (float)hw, part, ticket);
  return gg::launch_status("lpips_tail_fwd");
}

extern "C" int gg_lpips_tail_bwd_f32(float* dfeats, const float* feats, const float* lin, const float* grad_out, int n,
                                     int c, long long hw, float eps, int accumulate, void* stream) {
  if (n <= 0) return 0;
  if (!dfeats || !feats || !grad_out || c <= 0 || hw <= 0 || n > 65535)
    return gg::fail(-2, "lpips_tail_bwd: bad arguments");
  lpips_tail_bwd_kernel<<<tail_grid(n, hw), 256, 0, gg::as_stream(stream)>>>(dfeats, feats, lin, grad_out, n, c, hw, eps,
                                                                            1.f / (float)hw, accumulate);
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
