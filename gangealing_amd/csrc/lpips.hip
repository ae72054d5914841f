// LPIPS / SimCLR-VGG perceptual distance tail, one feature tap per call
// (reference models/losses/lpips.py:26-28 normalize_tensor, :190-199 diffs / lins / spatial_average):
//   u = f / (sqrt(sum_c f^2) + eps)                     per pixel, both images
//   out[n] = 1/(H*W) * sum_pixels sum_c lin[c] * (u0[c] - u1[c])^2      (lin = NULL: all ones = lpips=False branch)
// feats (2N, C, H, W): samples [0,N) are image 0, [N,2N) image 1 (one batched backbone pass).
// As torch ops this is ~10 element-wise / reduction passes over the feature tensors forward and ~20 backward;
// here the forward reads the features twice (norms, then the weighted squared difference - the second read hits the
// Infinity Cache) and the backward reads them twice and writes the gradient once.  HBM-bound; lanes run along the
// pixel axis (contiguous in NCHW), channels are a strided loop.
#include "../../include/gangealing_hip.h"
#include "gg_common.h"

namespace {

__global__ __launch_bounds__(256) void lpips_tail_fwd_kernel(float* __restrict__ out, const float* __restrict__ f,
                                                             const float* __restrict__ lin, int n, int c,
                                                             long long hw, float eps, float inv_hw) {
  __shared__ float red[4];
  const int s = blockIdx.y;                                    // sample
  const float* f0 = f + (size_t)s * c * hw;
  const float* f1 = f + (size_t)(s + n) * c * hw;
  float acc = 0.f;
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < hw; p += (long long)gridDim.x * 256) {
    float s00 = 0.f, s11 = 0.f;
    for (int k = 0; k < c; ++k) {
      const float a = f0[(size_t)k * hw + p], b = f1[(size_t)k * hw + p];
      s00 += a * a;
      s11 += b * b;
    }
    const float a0 = 1.f / (sqrtf(s00) + eps), a1 = 1.f / (sqrtf(s11) + eps);
    float d = 0.f;
    for (int k = 0; k < c; ++k) {
      const float t = f0[(size_t)k * hw + p] * a0 - f1[(size_t)k * hw + p] * a1;
      d += (lin ? lin[k] : 1.f) * t * t;
    }
    acc += d;
  }
  const float tot = gg::block_sum_256<float>(acc, red);
  if (threadIdx.x == 0) unsafeAtomicAdd(out + s, tot * inv_hw);
}

// df0[c] = a0 * q[c] - u0[c] * (sum_k q[k] u0[k]) / n0,   q[c] = 2 g/(HW) lin[c] (u0[c] - u1[c]);  df1 with -q.
__global__ __launch_bounds__(256) void lpips_tail_bwd_kernel(float* __restrict__ df, const float* __restrict__ f,
                                                             const float* __restrict__ lin,
                                                             const float* __restrict__ gout, int n, int c,
                                                             long long hw, float eps, float inv_hw) {
  const int s = blockIdx.y;
  const float* f0 = f + (size_t)s * c * hw;
  const float* f1 = f + (size_t)(s + n) * c * hw;
  float* d0 = df + (size_t)s * c * hw;
  float* d1 = df + (size_t)(s + n) * c * hw;
  const float g2 = 2.f * gout[s] * inv_hw;
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < hw; p += (long long)gridDim.x * 256) {
    float s00 = 0.f, s11 = 0.f;
    for (int k = 0; k < c; ++k) {
      const float a = f0[(size_t)k * hw + p], b = f1[(size_t)k * hw + p];
      s00 += a * a;
      s11 += b * b;
    }
    const float n0 = sqrtf(s00), n1 = sqrtf(s11);
    const float a0 = 1.f / (n0 + eps), a1 = 1.f / (n1 + eps);
    // t0 = sum_k q[k] u0[k], t1 = sum_k (-q[k]) u1[k]
    float t0 = 0.f, t1 = 0.f;
    for (int k = 0; k < c; ++k) {
      const float u0 = f0[(size_t)k * hw + p] * a0, u1 = f1[(size_t)k * hw + p] * a1;
      const float q = g2 * (lin ? lin[k] : 1.f) * (u0 - u1);
      t0 += q * u0;
      t1 -= q * u1;
    }
    const float r0 = n0 > 0.f ? t0 / n0 : 0.f, r1 = n1 > 0.f ? t1 / n1 : 0.f;
    for (int k = 0; k < c; ++k) {
      const float u0 = f0[(size_t)k * hw + p] * a0, u1 = f1[(size_t)k * hw + p] * a1;
      const float q = g2 * (lin ? lin[k] : 1.f) * (u0 - u1);
      d0[(size_t)k * hw + p] = a0 * q - u0 * r0;
      d1[(size_t)k * hw + p] = -a1 * q - u1 * r1;
    }
  }
}

dim3 tail_grid(int n, long long hw) {
  long long bx = (hw + 255) / 256;
  if (bx > 4096) bx = 4096;
  return dim3((unsigned)bx, (unsigned)n);
}

}  // namespace

extern "C" int gg_lpips_tail_fwd_f32(float* out, const float* feats, const float* lin, int n, int c, long long hw,
                                     float eps, void* stream) {
  if (n <= 0) return 0;
  if (!out || !feats || c <= 0 || hw <= 0 || n > 65535) return gg::fail(-2, "lpips_tail_fwd: bad arguments");
  hipStream_t st = gg::as_stream(stream);
  hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * n, st);
  if (e != hipSuccess) return gg::fail((int)e, "lpips_tail_fwd: memset failed");
  lpips_tail_fwd_kernel<<<tail_grid(n, hw), 256, 0, st>>>(out, feats, lin, n, c, hw, eps, 1.f / (float)hw);
  return gg::launch_status("lpips_tail_fwd");
}

extern "C" int gg_lpips_tail_bwd_f32(float* dfeats, const float* feats, const float* lin, const float* grad_out, int n,
                                     int c, long long hw, float eps, void* stream) {
  if (n <= 0) return 0;
  if (!dfeats || !feats || !grad_out || c <= 0 || hw <= 0 || n > 65535)
    return gg::fail(-2, "lpips_tail_bwd: bad arguments");
  lpips_tail_bwd_kernel<<<tail_grid(n, hw), 256, 0, gg::as_stream(stream)>>>(dfeats, feats, lin, grad_out, n, c, hw, eps,
                                                                            1.f / (float)hw);
  return gg::launch_status("lpips_tail_bwd");
}
