// a2  fused bias + leaky-relu, forward and one-pass backward (+ bias gradient).
// Pure HBM streaming: 8 B/elem forward, 12 B/elem backward.  16 B per lane loads/stores, grid
// capped at 256 CUs x 8 blocks with a grid-stride loop.
#include "../../include/gangealing_hip.h"
#include "gg_common.h"

namespace {

template <typename T>
__device__ __forceinline__ T act_apply(T v, T r, int mode, T alpha, T scale) {
  // mode = act*10 + grad  (reference fused_bias_act_kernel.cu:36-47)
  T y;
  switch (mode) {
    case 12: case 32: y = T(0); break;
    case 30: y = (v > T(0)) ? v : v * alpha; break;
    case 31: y = (r > T(0)) ? v : v * alpha; break;
    default: y = v; break;                       // 10, 11 and unknown -> linear
  }
  return y * scale;
}

template <typename T> struct Vec4 { T v[4]; };

// VEC = 4: every group of 4 consecutive elements shares one bias entry (step_b % 4 == 0).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void fused_bias_act_kernel(
    T* __restrict__ out, const T* __restrict__ x, const T* __restrict__ b, const T* __restrict__ ref,
    int mode, T alpha, T scale, long long n_items, long long step_b, int size_b) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const bool small = n_items * VEC < (1LL << 31) && step_b < (1LL << 31);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += stride) {
    const long long e = i * VEC;
    T bias = T(0);
    // (32-bit index math whenever the tensor allows it: a 64-bit division is ~100 VALU instructions per lane)
    if (b) bias = small ? b[((unsigned)e / (unsigned)step_b) % (unsigned)size_b] : b[(e / step_b) % size_b];
    if (VEC == 4) {
      Vec4<T> xv = *reinterpret_cast<const Vec4<T>*>(x + e);
      Vec4<T> rv;
      if (ref) rv = *reinterpret_cast<const Vec4<T>*>(ref + e);
      Vec4<T> yv;
#pragma unroll
      for (int j = 0; j < 4; ++j) yv.v[j] = act_apply<T>(xv.v[j] + bias, ref ? rv.v[j] : T(0), mode, alpha, scale);
      *reinterpret_cast<Vec4<T>*>(out + e) = yv;
    } else {
      out[e] = act_apply<T>(x[e] + bias, ref ? ref[e] : T(0), mode, alpha, scale);
    }
  }
}

// Backward: grid = (splits, C).  Block (s, c) walks hw-range s of every sample n for channel c,
// writes grad_in and its share of the channel's bias gradient; the channel's last block adds the shares in block
// order (gg::ordered_grid_sum: fixed summation order, no float atomics) and writes grad_bias[c].
template <typename T, int VEC>
__global__ __launch_bounds__(256) void fused_lrelu_bwd_kernel(
    T* __restrict__ gin, T* __restrict__ gbias, const T* __restrict__ gout, const T* __restrict__ outv,
    T alpha, T scale, int n, int c, long long hw, long long chunk, T* part, unsigned* ticket, int accumulate) {
  __shared__ T red[4];
  const int ch = blockIdx.y;
  const long long lo = (long long)blockIdx.x * chunk;
  long long hi = lo + chunk;
  if (hi > hw) hi = hw;
  T acc = T(0);
  // the block's work is the hw-range [lo, hi) of EVERY sample: one flat index over (sample, position) keeps all 256 lanes
  // busy on the small planes too (hw = 256: a per-sample loop ran 64 lanes for 16 dependent iterations; round 6)
  const long long per = (hi - lo + VEC - 1) / VEC;          // vectors per sample in this range
  const long long items = (long long)n * per;
  const bool small = items < (1LL << 31);
  for (long long i = threadIdx.x; i < items; i += 256) {
    long long s, q;
    if (small) { const unsigned ss = (unsigned)i / (unsigned)per; s = ss; q = (unsigned)i - ss * (unsigned)per; }
    else { s = i / per; q = i - s * per; }
    const long long p = lo + q * VEC;
    const long long base = (s * c + ch) * hw;
    if (VEC == 4) {
      Vec4<T> g = *reinterpret_cast<const Vec4<T>*>(gout + base + p);
      Vec4<T> o = *reinterpret_cast<const Vec4<T>*>(outv + base + p);
      Vec4<T> r;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r.v[j] = ((o.v[j] > T(0)) ? g.v[j] : g.v[j] * alpha) * scale;
        acc += r.v[j];
      }
      *reinterpret_cast<Vec4<T>*>(gin + base + p) = r;
    } else {
      const T g = gout[base + p], o = outv[base + p];
      const T r = ((o > T(0)) ? g : g * alpha) * scale;
      gin[base + p] = r;
      acc += r;
    }
  }
  if (gbias) {
    T v[1] = {gg::block_sum_256<T>(acc, red)};
    if (gg::ordered_grid_sum<T, 1>(v, part, ticket, ch, blockIdx.x, gridDim.x, red))
      gbias[ch] = (accumulate ? gbias[ch] : T(0)) + v[0];
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// binary16 tensors (the reference dispatches half, fused_bias_act_kernel.cu:89): fp32 arithmetic, one rounding per
// element.  VEC = 8: 16 bytes per lane, all 8 elements under one bias entry.
typedef _Float16 h16;
struct H8 { h16 v[8]; };
template <int VEC>
__global__ __launch_bounds__(256) void fused_bias_act_f16_kernel(h16* __restrict__ out, const h16* __restrict__ x,
                                                                 const h16* __restrict__ b, const h16* __restrict__ ref,
                                                                 int mode, float alpha, float scale, long long n_items,
                                                                 long long step_b, int size_b) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += stride) {
    const long long e = i * VEC;
    const float bias = b ? (float)b[(e / step_b) % size_b] : 0.f;
    if (VEC == 8) {
      const H8 xv = *reinterpret_cast<const H8*>(x + e);
      H8 rv;
      if (ref) rv = *reinterpret_cast<const H8*>(ref + e);
      H8 yv;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        yv.v[j] = (h16)act_apply<float>((float)xv.v[j] + bias, ref ? (float)rv.v[j] : 0.f, mode, alpha, scale);
      *reinterpret_cast<H8*>(out + e) = yv;
    } else {
      out[e] = (h16)act_apply<float>((float)x[e] + bias, ref ? (float)ref[e] : 0.f, mode, alpha, scale);
    }
  }
}

// grid = (splits, C) as fused_lrelu_bwd_kernel; the bias gradient is accumulated in fp32
__global__ __launch_bounds__(256) void fused_lrelu_bwd_f16_kernel(h16* __restrict__ gin, float* __restrict__ gbias,
                                                                  const h16* __restrict__ gout,
                                                                  const h16* __restrict__ outv, float alpha, float scale,
                                                                  int n, int c, long long hw, long long chunk,
                                                                  float* part, unsigned* ticket) {
  __shared__ float red[4];
  const int ch = blockIdx.y;
  const long long lo = (long long)blockIdx.x * chunk;
  long long hi = lo + chunk;
  if (hi > hw) hi = hw;
  float acc = 0.f;
  for (int s = 0; s < n; ++s) {
    const long long base = ((long long)s * c + ch) * hw;
    for (long long p = lo + threadIdx.x; p < hi; p += 256) {
      const float g = (float)gout[base + p], o = (float)outv[base + p];
      const h16 r = (h16)(((o > 0.f) ? g : g * alpha) * scale);
      gin[base + p] = r;
      acc += (float)r;                 // the reference sums the rounded grad_input (fused_act.py:33-38)
    }
  }
  if (gbias) {
    float v[1] = {gg::block_sum_256<float>(acc, red)};
    if (gg::ordered_grid_sum<float, 1>(v, part, ticket, ch, blockIdx.x, gridDim.x, red)) gbias[ch] = v[0];
  }
}

// StyledConv tail in one pass (networks.py:291-298,344-350): y = lrelu(x + nw * noise[n,0,hw] + b[c]) * scale.
// 12 B/elem of the big tensor instead of 3 passes (mul, add, activation).
__global__ __launch_bounds__(256) void noise_bias_act_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                             const float* __restrict__ noise,
                                                             const float* __restrict__ noise_weight,
                                                             const float* __restrict__ b, float alpha, float scale,
                                                             long long n_vec, int c, long long hw) {
  const float nw = noise_weight[0];
  const bool small = n_vec * 4 < (1LL << 31) && hw < (1LL << 31);
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
    const long long e = i * 4;
    long long plane, pix, n;                        // plane = n * c + ch
    if (small) {
      const unsigned pl = (unsigned)e / (unsigned)hw;
      plane = pl; pix = (unsigned)e - pl * (unsigned)hw; n = pl / (unsigned)c;
    } else {
      plane = e / hw; pix = e - plane * hw; n = plane / c;
    }
    const int ch = (int)(plane - n * c);
    const float4 xv = *reinterpret_cast<const float4*>(x + e);
    const float4 nv = *reinterpret_cast<const float4*>(noise + n * hw + pix);
    const float bias = b[ch];
    float4 y;
    y.x = act_apply<float>(xv.x + nw * nv.x + bias, 0.f, 30, alpha, scale);
    y.y = act_apply<float>(xv.y + nw * nv.y + bias, 0.f, 30, alpha, scale);
    y.z = act_apply<float>(xv.z + nw * nv.z + bias, 0.f, 30, alpha, scale);
    y.w = act_apply<float>(xv.w + nw * nv.w + bias, 0.f, 30, alpha, scale);
    *reinterpret_cast<float4*>(out + e) = y;
  }
}

template <typename T>
int fused_bias_act_impl(T* out, const T* x, const T* bias, const T* ref, int act, int grad, T alpha, T scale,
                        long long size_x, long long step_b, int size_b, void* stream) {
  if (size_x == 0) return 0;
  if (size_x < 0 || !out || !x) return gg::fail(-2, "fused_bias_act: bad arguments");
  if (bias && (step_b <= 0 || size_b <= 0)) return gg::fail(-2, "fused_bias_act: bias needs step_b, size_b > 0");
  const int mode = act * 10 + grad;
  const int vecn = 16 / (int)sizeof(T);            // elements per 16-byte access (4 for f32, 2 for f64)
  const bool vec = (vecn == 4) && (size_x % 4 == 0) && (!bias || step_b % 4 == 0) && aligned16(out) && aligned16(x) &&
                   (!ref || aligned16(ref));
  hipStream_t st = gg::as_stream(stream);
  if (vec) {
    const long long items = size_x / 4;
    fused_bias_act_kernel<T, 4><<<gg::stream_grid(items, 256), 256, 0, st>>>(out, x, bias, ref, mode, alpha, scale,
                                                                             items, step_b, size_b);
  } else {
    fused_bias_act_kernel<T, 1><<<gg::stream_grid(size_x, 256), 256, 0, st>>>(out, x, bias, ref, mode, alpha, scale,
                                                                              size_x, step_b, size_b);
  }
  return gg::launch_status("fused_bias_act");
}

template <typename T>
int fused_lrelu_bwd_impl(T* gin, T* gbias, const T* gout, const T* outv, T alpha, T scale, int n, int c,
                         long long hw, void* stream, int accumulate = 0) {
  if (n <= 0 || c <= 0 || hw <= 0) return 0;
  if (!gin || !gout || !outv) return gg::fail(-2, "fused_lrelu_bwd: null pointer");
  if (c > 65535) return gg::fail(-2, "fused_lrelu_bwd: more than 65535 channels");
  hipStream_t st = gg::as_stream(stream);
  const bool vec = sizeof(T) == 4 && (hw % 4 == 0) && aligned16(gin) && aligned16(gout) && aligned16(outv);
  const int per = vec ? 4 : 1;
  // enough blocks to fill the chip (>= ~2048) but at least one full 256-thread sweep per block
  long long want = (2048 + c - 1) / c;
  long long max_splits = (hw + 256LL * per - 1) / (256LL * per);
  long long splits = want < 1 ? 1 : want;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  long long chunk = (hw + splits - 1) / splits;
  chunk = (chunk + per - 1) / per * per;
  splits = (hw + chunk - 1) / chunk;
  if (c > gg::kTickets) { splits = 1; chunk = hw; }        // one ticket per channel
  T* part = nullptr;
  unsigned* ticket = nullptr;
  if (gbias && splits > 1) {
    part = reinterpret_cast<T*>(gg::scratch(st, sizeof(T) * (size_t)c * splits));
    ticket = gg::tickets(st);
    if (!part || !ticket) return -3;
  }
  dim3 grid((unsigned)splits, (unsigned)c);
  if (vec)
    fused_lrelu_bwd_kernel<T, 4><<<grid, 256, 0, st>>>(gin, gbias, gout, outv, alpha, scale, n, c, hw, chunk, part, ticket,
                                                       accumulate);
  else
    fused_lrelu_bwd_kernel<T, 1><<<grid, 256, 0, st>>>(gin, gbias, gout, outv, alpha, scale, n, c, hw, chunk, part, ticket,
                                                       accumulate);
  return gg::launch_status("fused_lrelu_bwd");
}

}  // namespace

extern "C" int gg_fused_bias_act_f32(float* out, const float* x, const float* bias, const float* ref, int act,
                                     int grad, float alpha, float scale, long long size_x, long long step_b,
                                     int size_b, void* stream) {
  return fused_bias_act_impl<float>(out, x, bias, ref, act, grad, alpha, scale, size_x, step_b, size_b, stream);
}
extern "C" int gg_fused_bias_act_f64(double* out, const double* x, const double* bias, const double* ref, int act,
                                     int grad, double alpha, double scale, long long size_x, long long step_b,
                                     int size_b, void* stream) {
  return fused_bias_act_impl<double>(out, x, bias, ref, act, grad, alpha, scale, size_x, step_b, size_b, stream);
}
extern "C" int gg_fused_bias_act_f16(unsigned short* out, const unsigned short* x, const unsigned short* bias,
                                     const unsigned short* ref, int act, int grad, float alpha, float scale,
                                     long long size_x, long long step_b, int size_b, void* stream) {
  if (size_x == 0) return 0;
  if (size_x < 0 || !out || !x) return gg::fail(-2, "fused_bias_act: bad arguments");
  if (bias && (step_b <= 0 || size_b <= 0)) return gg::fail(-2, "fused_bias_act: bias needs step_b, size_b > 0");
  const int mode = act * 10 + grad;
  const bool vec = (size_x % 8 == 0) && (!bias || step_b % 8 == 0) && aligned16(out) && aligned16(x) &&
                   (!ref || aligned16(ref));
  hipStream_t st = gg::as_stream(stream);
  h16* o = reinterpret_cast<h16*>(out);
  const h16 *xi = reinterpret_cast<const h16*>(x), *bi = reinterpret_cast<const h16*>(bias),
            *ri = reinterpret_cast<const h16*>(ref);
  if (vec)
    fused_bias_act_f16_kernel<8><<<gg::stream_grid(size_x / 8, 256), 256, 0, st>>>(o, xi, bi, ri, mode, alpha, scale,
                                                                                   size_x / 8, step_b, size_b);
  else
    fused_bias_act_f16_kernel<1><<<gg::stream_grid(size_x, 256), 256, 0, st>>>(o, xi, bi, ri, mode, alpha, scale, size_x,
                                                                               step_b, size_b);
  return gg::launch_status("fused_bias_act");
}
extern "C" int gg_fused_lrelu_bwd_f16(unsigned short* grad_in, float* grad_bias, const unsigned short* grad_out,
                                      const unsigned short* out, float alpha, float scale, int n, int c, long long hw,
                                      void* stream) {
  if (n <= 0 || c <= 0 || hw <= 0) return 0;
  if (!grad_in || !grad_out || !out) return gg::fail(-2, "fused_lrelu_bwd: null pointer");
  if (c > 65535) return gg::fail(-2, "fused_lrelu_bwd: more than 65535 channels");
  hipStream_t st = gg::as_stream(stream);
  long long splits = (2048 + c - 1) / c;
  const long long max_splits = (hw + 255) / 256;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  if (c > gg::kTickets) splits = 1;
  const long long chunk = (hw + splits - 1) / splits;
  splits = (hw + chunk - 1) / chunk;
  float* part = nullptr;
  unsigned* ticket = nullptr;
  if (grad_bias && splits > 1) {
    part = reinterpret_cast<float*>(gg::scratch(st, sizeof(float) * (size_t)c * splits));
    ticket = gg::tickets(st);
    if (!part || !ticket) return -3;
  }
  dim3 grid((unsigned)splits, (unsigned)c);
  fused_lrelu_bwd_f16_kernel<<<grid, 256, 0, st>>>(reinterpret_cast<h16*>(grad_in), grad_bias,
                                                   reinterpret_cast<const h16*>(grad_out),
                                                   reinterpret_cast<const h16*>(out), alpha, scale, n, c, hw, chunk,
                                                   part, ticket);
  return gg::launch_status("fused_lrelu_bwd");
}
extern "C" int gg_noise_bias_act_f32(float* out, const float* x, const float* noise, const float* noise_weight,
                                     const float* bias, float alpha, float scale, int n, int c, long long hw,
                                     void* stream) {
  const long long total = (long long)n * c * hw;
  if (total <= 0) return 0;
  if (!out || !x || !noise || !noise_weight || !bias) return gg::fail(-2, "noise_bias_act: null pointer");
  if (hw % 4 != 0 || !aligned16(out) || !aligned16(x) || !aligned16(noise))
    return gg::fail(-2, "noise_bias_act: hw must be a multiple of 4 and pointers 16-byte aligned");
  noise_bias_act_kernel<<<gg::stream_grid(total / 4, 256), 256, 0, gg::as_stream(stream)>>>(
      out, x, noise, noise_weight, bias, alpha, scale, total / 4, c, hw);
  return gg::launch_status("noise_bias_act");
}

extern "C" int gg_fused_lrelu_bwd_f32(float* grad_in, float* grad_bias, const float* grad_out, const float* out,
                                      float alpha, float scale, int n, int c, long long hw, void* stream) {
  return fused_lrelu_bwd_impl<float>(grad_in, grad_bias, grad_out, out, alpha, scale, n, c, hw, stream);
}
extern "C" int gg_fused_lrelu_bwd_acc_f32(float* grad_in, float* grad_bias, const float* grad_out, const float* out,
                                          float alpha, float scale, int n, int c, long long hw, int accumulate,
                                          void* stream) {
  return fused_lrelu_bwd_impl<float>(grad_in, grad_bias, grad_out, out, alpha, scale, n, c, hw, stream, accumulate);
}
extern "C" int gg_fused_lrelu_bwd_f64(double* grad_in, double* grad_bias, const double* grad_out, const double* out,
                                      double alpha, double scale, int n, int c, long long hw, void* stream) {
  return fused_lrelu_bwd_impl<double>(grad_in, grad_bias, grad_out, out, alpha, scale, n, c, hw, stream);
}

// (a + b) * scale in one pass: the residual merge of ResBlock, (out + skip) / sqrt(2) (networks.py:392-393).
namespace {
__global__ __launch_bounds__(256) void add_scale_kernel(float* __restrict__ out, const float* __restrict__ a,
                                                        const float* __restrict__ b, float scale, long long n4,
                                                        long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
    float4 r;
    r.x = (x.x + y.x) * scale; r.y = (x.y + y.y) * scale; r.z = (x.z + y.z) * scale; r.w = (x.w + y.w) * scale;
    reinterpret_cast<float4*>(out)[i] = r;
  }
  for (long long i = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = (a[i] + b[i]) * scale;
}
}  // namespace

extern "C" int gg_add_scale_f32(float* out, const float* a, const float* b, float scale, long long n, void* stream) {
  if (n <= 0) return 0;
  if (!out || !a || !b) return gg::fail(-2, "add_scale: null pointer");
  const bool vec = aligned16(out) && aligned16(a) && aligned16(b);
  const long long n4 = vec ? n / 4 : 0;
  add_scale_kernel<<<gg::stream_grid(vec ? n4 + 1 : n, 256), 256, 0, gg::as_stream(stream)>>>(out, a, b, scale, n4, n);
  return gg::launch_status("add_scale");
}

