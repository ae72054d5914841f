// Style modulation of one ModulatedConv2d layer in one C call (reference models/stylegan2/networks.py:214-216
// EqualLinear, :244-249 style/demodulation):
//   style[n,ci] = sum_k latent[n,k] * W[ci,k] * w_scale + b[ci] * b_scale          (EqualLinear)
//   demod[n,co] = rsqrt(sum_ci style[n,ci]^2 * wsq[co,ci] + eps),  wsq[co,ci] = sum_taps (scale*Wconv)^2
// The shared-weight form of the modulated convolution (csrc/conv_mfma.hip) needs exactly these two small
// per-sample vectors; as separate torch ops they were seven ~5 us launches per layer and per generator pass.
// Two launches of one row-dot kernel (style, then demod): a wave owns one weight row, keeps it in registers and
// dots it with every sample of the batch (staged in LDS), so each weight matrix is read once per launch.
#include "../../include/gangealing_hip.h"
#include "gg_common.h"

namespace {

constexpr int LDS_FLOATS = 16384;      // batch chunk x reduction length staged per block
constexpr int MAX_DIM = 2048;          // reduction length (32 row elements per lane)

// out[n, r] = f(scale * sum_k g(in[n*in_stride + k]) * m[r, k] + bias[r] * bias_scale)
//   SQUARE: g(v) = v*v else v;  RSQRT: f(v) = rsqrt(v + eps) else v
template <bool SQUARE, bool RSQRT, int KPL>      // KPL: row elements per lane (reduction length <= 64 * KPL)
__global__ __launch_bounds__(256) void rowdot_kernel(float* __restrict__ out, const float* __restrict__ in,
                                                     long long in_stride, const float* __restrict__ m,
                                                     const float* __restrict__ bias, int n_total, int nb, int kdim,
                                                     int rows, float scale, float bias_scale, float eps) {
  __shared__ float sin[LDS_FLOATS];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int n0 = blockIdx.y * nb;
  const int ncount = min(nb, n_total - n0);
  // the weight row first (longest latency), then the batch chunk
  const int r = blockIdx.x * 4 + wid;
  const float* row = m + (size_t)(r < rows ? r : 0) * kdim;
  float wreg[KPL];
#pragma unroll
  for (int j = 0; j < KPL; ++j) {
    const int k = lane + 64 * j;
    wreg[j] = k < kdim ? row[k] : 0.f;
  }
  // stage the batch chunk: 8 independent loads in flight per thread (a load -> store loop per sample exposed one
  // memory round trip per iteration: 17 us for a 5 us kernel)
  const int total = ncount * kdim;
  for (int base = 0; base < total; base += 256 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * 256 + threadIdx.x;
      const int n = i / kdim, k = i - n * kdim;
      v[u] = i < total ? in[(size_t)(n0 + n) * in_stride + k] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * 256 + threadIdx.x;
      if (i < total) sin[i] = SQUARE ? v[u] * v[u] : v[u];
    }
  }
  __syncthreads();
  if (r >= rows) return;
  const float bb = bias ? bias[r] * bias_scale : 0.f;
  for (int nb4 = 0; nb4 < ncount; nb4 += 4) {           // four samples at a time: independent shuffle chains
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* x = sin + (nb4 + u < ncount ? nb4 + u : nb4) * kdim;
#pragma unroll
      for (int j = 0; j < KPL; ++j) {
        const int k = lane + 64 * j;
        acc[u] += (k < kdim ? x[k] : 0.f) * wreg[j];
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] += __shfl_down(acc[u], off, 64);
    if (lane == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (nb4 + u < ncount) {
          const float v = acc[u] * scale + bb;
          out[(size_t)(n0 + nb4 + u) * rows + r] = RSQRT ? rsqrtf(v + eps) : v;
        }
    }
  }
}

template <bool SQUARE, bool RSQRT>
void launch_rowdot(dim3 grid, hipStream_t st, float* out, const float* in, long long in_stride, const float* m,
                   const float* bias, int n, int nb, int kdim, int rows, float scale, float bias_scale, float eps) {
  if (kdim <= 512)
    rowdot_kernel<SQUARE, RSQRT, 8><<<grid, 256, 0, st>>>(out, in, in_stride, m, bias, n, nb, kdim, rows, scale,
                                                          bias_scale, eps);
  else if (kdim <= 1024)
    rowdot_kernel<SQUARE, RSQRT, 16><<<grid, 256, 0, st>>>(out, in, in_stride, m, bias, n, nb, kdim, rows, scale,
                                                           bias_scale, eps);
  else
    rowdot_kernel<SQUARE, RSQRT, 32><<<grid, 256, 0, st>>>(out, in, in_stride, m, bias, n, nb, kdim, rows, scale,
                                                           bias_scale, eps);
}

}  // namespace

namespace {
// g[n,c,p] += sum_{k<3} (w[k,c] * wscale * style[n,c]) * r[n,k,p]: the data gradient of a modulated 1x1 "ToRGB"
// convolution (networks.py:352-372, demodulate = False) added straight into the gradient that the other consumer of
// the same activation produced.  Replaces: ToRGB dgrad conv (writes a full-size tensor) + autograd's 3-pass add.
__global__ __launch_bounds__(256) void torgb_dgrad_add_kernel(float* __restrict__ g, const float* __restrict__ r,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ style, float wscale, int c,
                                                              long long hw4) {
  const int plane = blockIdx.y;                        // n * c + ch
  const int n = plane / c, ch = plane - n * c;
  const float s = style[(size_t)n * c + ch] * wscale;
  const float m0 = w[ch] * s, m1 = w[c + ch] * s, m2 = w[2 * c + ch] * s;
  float4* gp = reinterpret_cast<float4*>(g + (size_t)plane * hw4 * 4);
  const float4* r0 = reinterpret_cast<const float4*>(r + ((size_t)n * 3 + 0) * hw4 * 4);
  const float4* r1 = reinterpret_cast<const float4*>(r + ((size_t)n * 3 + 1) * hw4 * 4);
  const float4* r2 = reinterpret_cast<const float4*>(r + ((size_t)n * 3 + 2) * hw4 * 4);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < hw4; i += (long long)gridDim.x * 256) {
    float4 v = gp[i];
    const float4 a = r0[i], b = r1[i], d = r2[i];
    v.x += m0 * a.x + m1 * b.x + m2 * d.x;
    v.y += m0 * a.y + m1 * b.y + m2 * d.y;
    v.z += m0 * a.z + m1 * b.z + m2 * d.z;
    v.w += m0 * a.w + m1 * b.w + m2 * d.w;
    gp[i] = v;
  }
}
}  // namespace

extern "C" int gg_torgb_dgrad_add_f32(float* g, const float* grad_rgb, const float* w, const float* style, float wscale,
                                      int n, int c, long long hw, void* stream) {
  if (n <= 0 || c <= 0 || hw <= 0) return 0;
  if (!g || !grad_rgb || !w || !style) return gg::fail(-2, "torgb_dgrad_add: null pointer");
  if (hw % 4 != 0 || (reinterpret_cast<uintptr_t>(g) & 15) || (reinterpret_cast<uintptr_t>(grad_rgb) & 15))
    return gg::fail(-2, "torgb_dgrad_add: hw must be a multiple of 4 and g / grad_rgb 16-byte aligned");
  long long bx = (hw / 4 + 255) / 256;
  if (bx > 64) bx = 64;
  dim3 grid((unsigned)bx, (unsigned)(n * c));
  if ((long long)n * c > 65535) return gg::fail(-2, "torgb_dgrad_add: more than 65535 planes");
  torgb_dgrad_add_kernel<<<grid, 256, 0, gg::as_stream(stream)>>>(g, grad_rgb, w, style, wscale, c, hw / 4);
  return gg::launch_status("torgb_dgrad_add");
}

// ---------------------------------------------------------------------------------------------------------------
// Style bank: the modulation vectors of MANY layers in two launches (all styles, then all demodulations) instead of
// two launches per layer.  The generator is frozen, so the job table (weights, sizes, output offsets) is built once;
// per call only the latent base pointer and the output buffer change.
//   phase 0 job: out[n, r] = sum_k latent[n, slot, k] * m[r, k] * scale + bias[r] * bias_scale        (EqualLinear)
//   phase 1 job: out[n, r] = rsqrt(sum_k in[n, k]^2 * m[r, k] + eps),  in = a phase-0 output         (demodulation)
struct StyleJob {
  const float* m;          // (rows, kdim) weight / squared-weight table
  const float* bias;       // (rows) or null
  long long out_off;       // float offset of this job's (n, rows) output inside `out_base`
  long long in_off;        // phase 0: latent slot index; phase 1: float offset of the (n, kdim) input inside `out_base`
  int kdim, rows;
  float scale, bias_scale, eps;
  int pad;
};
static_assert(sizeof(StyleJob) == 56, "StyleJob layout is mirrored by numpy in op/conv_mfma.py");

template <bool DEMOD>
__global__ __launch_bounds__(256) void style_bank_kernel(float* __restrict__ out_base, const float* __restrict__ latent,
                                                         long long lat_sample_stride, int slot_stride,
                                                         const StyleJob* __restrict__ jobs, int n_total) {
  __shared__ float sin[LDS_FLOATS];
  const StyleJob job = jobs[blockIdx.z];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wid;
  if (blockIdx.x * 4 >= job.rows) return;                  // whole block beyond this job's rows
  const int kdim = job.kdim;
  const int nb = min(LDS_FLOATS / kdim, n_total);
  const float* in = DEMOD ? out_base + job.in_off : latent + job.in_off * slot_stride;
  const long long in_stride = DEMOD ? kdim : lat_sample_stride;
  const float* row = job.m + (size_t)(r < job.rows ? r : 0) * kdim;
  float wreg[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = lane + 64 * j;
    wreg[j] = k < kdim ? row[k] : 0.f;
  }
  const float bb = (!DEMOD && job.bias && r < job.rows) ? job.bias[r] * job.bias_scale : 0.f;
  float* out = out_base + job.out_off;
  for (int n0 = 0; n0 < n_total; n0 += nb) {
    const int ncount = min(nb, n_total - n0);
    const int total = ncount * kdim;
    __syncthreads();
    for (int base = 0; base < total; base += 256 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * 256 + threadIdx.x;
        const int n = i / kdim, k = i - n * kdim;
        v[u] = i < total ? in[(size_t)(n0 + n) * in_stride + k] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * 256 + threadIdx.x;
        if (i < total) sin[i] = DEMOD ? v[u] * v[u] : v[u];
      }
    }
    __syncthreads();
    if (r < job.rows)
      for (int nb4 = 0; nb4 < ncount; nb4 += 4) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* x = sin + (nb4 + u < ncount ? nb4 + u : nb4) * kdim;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int k = lane + 64 * j;
            acc[u] += (k < kdim ? x[k] : 0.f) * wreg[j];
          }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[u] += __shfl_down(acc[u], off, 64);
        if (lane == 0) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (nb4 + u < ncount) {
              const float v = acc[u] * job.scale + bb;
              out[(size_t)(n0 + nb4 + u) * job.rows + r] = DEMOD ? rsqrtf(v + job.eps) : v;
            }
        }
      }
  }
}

extern "C" int gg_style_bank_f32(float* out_base, const float* latent, long long lat_sample_stride, int slot_stride,
                                 const void* style_jobs, int n_style_jobs, int max_style_rows, const void* demod_jobs,
                                 int n_demod_jobs, int max_demod_rows, int n, void* stream) {
  if (n <= 0 || n_style_jobs <= 0) return 0;
  if (!out_base || !latent || !style_jobs || (n_demod_jobs > 0 && !demod_jobs))
    return gg::fail(-2, "style_bank: null pointer");
  if (n_style_jobs > 65535 || n_demod_jobs > 65535) return gg::fail(-2, "style_bank: too many jobs");
  hipStream_t st = gg::as_stream(stream);
  style_bank_kernel<false><<<dim3((unsigned)((max_style_rows + 3) / 4), 1, (unsigned)n_style_jobs), 256, 0, st>>>(
      out_base, latent, lat_sample_stride, slot_stride, static_cast<const StyleJob*>(style_jobs), n);
  int rc = gg::launch_status("style_bank (styles)");
  if (rc || n_demod_jobs <= 0) return rc;
  style_bank_kernel<true><<<dim3((unsigned)((max_demod_rows + 3) / 4), 1, (unsigned)n_demod_jobs), 256, 0, st>>>(
      out_base, latent, lat_sample_stride, slot_stride, static_cast<const StyleJob*>(demod_jobs), n);
  return gg::launch_status("style_bank (demodulation)");
}

namespace {
// dstyle[n, ci] = dot_x[n, ci] + 2 style[n, ci] * sum_co t[n, co] * wsq[co, ci],  t = -0.5 * dot_y * demod^2
// (the style gradient of the shared-weight modulated convolution: dot_x = <d(style-scaled input), x> per plane, and
//  the demodulation branch d demod / d style = -demod^3 * style * wsq).
constexpr int STYLE_GRAD_MAX_COUT = 1024;
// block = 64 input channels x 4 waves; wave g sums the output channels co = g, g + 4, ... (eight independent partial sums
// per lane so that the wsq loads - one L2 round trip each - overlap), the waves meet in LDS.  grid = (ceil(cin / 64), n)
__global__ __launch_bounds__(256) void style_grad_kernel(float* __restrict__ dstyle, const float* __restrict__ dot_x,
                                                         const float* __restrict__ dot_y,
                                                         const float* __restrict__ demod,
                                                         const float* __restrict__ style,
                                                         const float* __restrict__ wsq, int cin, int cout) {
  __shared__ float t[STYLE_GRAD_MAX_COUT];
  __shared__ float red[4][64];
  const int n = blockIdx.y, lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int ci = blockIdx.x * 64 + lane;
  for (int co = threadIdx.x; co < cout; co += 256) {
    const float d = demod[(size_t)n * cout + co];
    t[co] = -0.5f * dot_y[(size_t)n * cout + co] * d * d;
  }
  __syncthreads();
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (ci < cin) {
    int co = g;
    for (; co + 28 < cout; co += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += t[co + 4 * u] * wsq[(size_t)(co + 4 * u) * cin + ci];
    }
    for (; co < cout; co += 4) acc[0] += t[co] * wsq[(size_t)co * cin + ci];
  }
  red[g][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (g == 0 && ci < cin) {
    const float sum = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    dstyle[(size_t)n * cin + ci] = dot_x[(size_t)n * cin + ci] + 2.f * style[(size_t)n * cin + ci] * sum;
  }
}
}  // namespace

extern "C" int gg_modconv_style_grad_f32(float* dstyle, const float* dot_x, const float* dot_y, const float* demod,
                                         const float* style, const float* wsq, int n, int cin, int cout,
                                         void* stream) {
  if (n <= 0 || cin <= 0) return 0;
  if (!dstyle || !dot_x || !dot_y || !demod || !style || !wsq || cout <= 0 || cout > STYLE_GRAD_MAX_COUT || n > 65535)
    return gg::fail(-2, "modconv_style_grad: bad arguments (cout <= %d)", STYLE_GRAD_MAX_COUT);
  style_grad_kernel<<<dim3((unsigned)((cin + 63) / 64), (unsigned)n), 256, 0, gg::as_stream(stream)>>>(
      dstyle, dot_x, dot_y, demod, style, wsq, cin, cout);
  return gg::launch_status("modconv_style_grad");
}

extern "C" int gg_style_demod_f32(float* style, float* demod, const float* latent, long long lat_stride,
                                  const float* w, const float* b, const float* wsq, int n, int style_dim, int cin,
                                  int cout, float w_scale, float b_scale, float eps, void* stream) {
  if (n <= 0 || cin <= 0) return 0;
  if (!style || !latent || !w) return gg::fail(-2, "style_demod: null pointer");
  if (demod && (!wsq || cout <= 0)) return gg::fail(-2, "style_demod: demodulation needs wsq and cout");
  if (style_dim <= 0 || style_dim > MAX_DIM || cin > MAX_DIM)
    return gg::fail(-2, "style_demod: style_dim and cin must be in [1, %d]", MAX_DIM);
  hipStream_t st = gg::as_stream(stream);
  {
    const int nb = LDS_FLOATS / style_dim < n ? LDS_FLOATS / style_dim : n;
    dim3 grid((unsigned)((cin + 3) / 4), (unsigned)((n + nb - 1) / nb));
    launch_rowdot<false, false>(grid, st, style, latent, lat_stride, w, b, n, nb, style_dim, cin, w_scale, b_scale, 0.f);
    const int rc = gg::launch_status("style_demod (style)");
    if (rc || !demod) return rc;
  }
  const int nb = LDS_FLOATS / cin < n ? LDS_FLOATS / cin : n;
  dim3 grid((unsigned)((cout + 3) / 4), (unsigned)((n + nb - 1) / nb));
  launch_rowdot<true, true>(grid, st, demod, style, cin, wsq, nullptr, n, nb, cin, cout, 1.f, 0.f, eps);
  return gg::launch_status("style_demod (demod)");
}
