// a1  upfirdn2d: zero-stuff by `up`, pad/crop, FIR with the flipped taps, decimate by `down`.
//
// Index math (derived from the definition, matches upfirdn2d_kernel.cu:149-206): for output o,
//   mid = o*down + up - 1 - pad0,  i0 = floor(mid / up),  t0 = i0*up + pad0 - o*down
//   out[o] = sum_j x[i0 + j] * kflip[t0 + j*up]      (x zero outside [0, in)), t0 + j*up < k
// with kflip the spatially flipped kernel.
//
// Two kernels:
//   * upfirdn2d_blur4_tile  - the hot case (up = down = 1, 4x4 taps: every Blur on the G / STN
//     path and its backward).  HBM-bound: a 64x64 output tile per 256-thread block, the (67x67)
//     input tile staged once in LDS (row reads by a wave are 64 consecutive words -> conflict
//     free), each lane owns one output column and slides a 4-row register window down 16 rows,
//     so every LDS word is read 4x instead of 16x and every output row is stored as one
//     coalesced 256 B wave row.  Tiles are XCD-remapped so neighbours share an L2.
//   * upfirdn2d_direct      - everything else (up/down = 2 on the 3-channel ToRGB skip, generic
//     kernels, tiny planes): one output per thread, taps staged in LDS.
#include "../../include/gangealing_hip.h"
#include "gg_common.h"

namespace {

constexpr int TILE = 64;          // output tile edge of the blur kernel
constexpr int TIN = TILE + 3;     // input tile edge (4 taps)
constexpr int ROWS_PER_WAVE = TILE / 4;

// Optional fusions for the generator's up-sampling StyledConv (networks.py:268-298, 344-350):
//   EPI: the blur is followed by NoiseInjection + FusedLeakyReLU -> applied to the result before the store;
//   PRO: the blur's INPUT is a leaky-ReLU gradient g * (ref > 0 ? 1 : alpha) * gain -> applied while the tile is
//        staged (the backward of an EPI blur is a PRO blur of the incoming gradient with ref = the saved output).
struct BlurFuse {
  const float* noise;      // (N, 1, out_h, out_w)
  const float* noise_w;    // device scalar
  const float* bias;       // (C)
  const float* ref;        // same shape as `in`
  float alpha, gain;
  int channels;
};

template <bool EPI, bool PRO>
__global__ __launch_bounds__(256) void upfirdn2d_blur4_tile(
    float* __restrict__ out, const float* __restrict__ in, const float* __restrict__ kernel,
    int planes, int in_h, int in_w, int out_h, int out_w, int pad_x0, int pad_y0,
    int tiles_x, int tiles_y, unsigned ntiles, const BlurFuse f) {
  __shared__ float sx[TIN][TIN + 1];
  __shared__ float sk[16];

  const unsigned logical = gg::xcd_remap(blockIdx.x, ntiles);
  const int tiles_per_plane = tiles_x * tiles_y;
  const int plane = logical / tiles_per_plane;
  const int t = logical - plane * tiles_per_plane;
  const int ty = t / tiles_x, tx = t - ty * tiles_x;
  const int oy0 = ty * TILE, ox0 = tx * TILE;
  const int iy0 = oy0 - pad_y0, ix0 = ox0 - pad_x0;       // input coords of the tile's first row/col

  if (threadIdx.x < 16) {
    const int ky = threadIdx.x >> 2, kx = threadIdx.x & 3;
    sk[threadIdx.x] = kernel[(3 - ky) * 4 + (3 - kx)];     // flipped taps
  }
  const float* src = in + (size_t)plane * in_h * in_w;
  for (int idx = threadIdx.x; idx < TIN * TIN; idx += 256) {
    const int r = idx / TIN, c = idx - r * TIN;
    const int iy = iy0 + r, ix = ix0 + c;
    float v = 0.f;
    if (iy >= 0 && iy < in_h && ix >= 0 && ix < in_w) {
      v = src[(size_t)iy * in_w + ix];
      if (PRO) v *= (f.ref[(size_t)plane * in_h * in_w + (size_t)iy * in_w + ix] > 0.f) ? f.gain : f.gain * f.alpha;
    }
    sx[r][c] = v;
  }
  __syncthreads();

  float kf[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) kf[i] = sk[i];

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r0 = wid * ROWS_PER_WAVE;
  const int ox = ox0 + lane;
  float* dst = out + (size_t)plane * out_h * out_w;
  const float* nz = nullptr;
  float nw = 0.f, ab = 0.f;
  if (EPI) {
    const int n = plane / f.channels;
    nz = f.noise + (size_t)n * out_h * out_w;
    nw = f.noise_w[0];
    ab = f.bias[plane - n * f.channels];
  }
  // Separable taps (every Blur of the networks: make_kernel is an outer product, networks.py:34-41): 4 horizontal +
  // 4 vertical FMAs per output and a 4-value register window instead of 16 + 16.  The kernel is VALU-issue bound
  // (~50 instructions per output in the 16-tap form), not bandwidth bound.  k[i][j] = a[i] * b[j] is checked on the
  // taps themselves; anything else takes the general path below.
  bool sep = kf[0] != 0.f;
  float ka[4], kb[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) kb[c] = kf[c];
#pragma unroll
  for (int r = 0; r < 4; ++r) ka[r] = kf[r * 4] / kf[0];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) sep = sep & (fabsf(ka[r] * kb[c] - kf[r * 4 + c]) <= 1e-6f * fabsf(kf[r * 4 + c]));
  if (sep) {
    float h0, h1, h2;
    {
      const float* p = &sx[r0][lane];
      h0 = p[0] * kb[0] + p[1] * kb[1] + p[2] * kb[2] + p[3] * kb[3];
      p += TIN + 1;
      h1 = p[0] * kb[0] + p[1] * kb[1] + p[2] * kb[2] + p[3] * kb[3];
      p += TIN + 1;
      h2 = p[0] * kb[0] + p[1] * kb[1] + p[2] * kb[2] + p[3] * kb[3];
    }
#pragma unroll
    for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
      const float* p = &sx[r0 + rr + 3][lane];
      const float h3 = p[0] * kb[0] + p[1] * kb[1] + p[2] * kb[2] + p[3] * kb[3];
      float acc = h0 * ka[0] + h1 * ka[1] + h2 * ka[2] + h3 * ka[3];
      const int oy = oy0 + r0 + rr;
      if (oy < out_h && ox < out_w) {
        if (EPI) {
          const float t = acc + nw * nz[(size_t)oy * out_w + ox] + ab;
          acc = (t > 0.f ? t : t * f.alpha) * f.gain;
        }
        dst[(size_t)oy * out_w + ox] = acc;
      }
      h0 = h1; h1 = h2; h2 = h3;
    }
    return;
  }
  float w0[4], w1[4], w2[4], w3[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    w0[c] = sx[r0 + 0][lane + c];
    w1[c] = sx[r0 + 1][lane + c];
    w2[c] = sx[r0 + 2][lane + c];
  }
#pragma unroll
  for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
#pragma unroll
    for (int c = 0; c < 4; ++c) w3[c] = sx[r0 + rr + 3][lane + c];
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) acc += w0[c] * kf[c];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc += w1[c] * kf[4 + c];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc += w2[c] * kf[8 + c];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc += w3[c] * kf[12 + c];
    const int oy = oy0 + r0 + rr;
    if (oy < out_h && ox < out_w) {
      if (EPI) {
        const float t = acc + nw * nz[(size_t)oy * out_w + ox] + ab;
        acc = (t > 0.f ? t : t * f.alpha) * f.gain;
      }
      dst[(size_t)oy * out_w + ox] = acc;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) { w0[c] = w1[c]; w1[c] = w2[c]; w2[c] = w3[c]; }
  }
}

constexpr int MAX_LDS_TAPS = 1024;

template <typename T>
__global__ __launch_bounds__(256) void upfirdn2d_direct(
    T* __restrict__ out, const T* __restrict__ in, const T* __restrict__ kernel,
    long long total, int in_h, int in_w, int out_h, int out_w, int kh, int kw,
    int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_y0) {
  __shared__ T sk[MAX_LDS_TAPS];
  const bool lds_taps = kh * kw <= MAX_LDS_TAPS;
  if (lds_taps) {
    for (int i = threadIdx.x; i < kh * kw; i += 256) {
      const int ky = i / kw, kx = i - ky * kw;
      sk[i] = kernel[(kh - 1 - ky) * kw + (kw - 1 - kx)];
    }
    __syncthreads();
  }
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const int ox = (int)(o % out_w);
    const long long q = o / out_w;
    const int oy = (int)(q % out_h);
    const long long plane = q / out_h;
    const int mid_x = ox * down_x + up_x - 1 - pad_x0;
    const int mid_y = oy * down_y + up_y - 1 - pad_y0;
    const int ix0 = gg::floor_div(mid_x, up_x), iy0 = gg::floor_div(mid_y, up_y);
    const int tx0 = ix0 * up_x + pad_x0 - ox * down_x;
    const int ty0 = iy0 * up_y + pad_y0 - oy * down_y;
    const T* src = in + (size_t)plane * in_h * in_w;
    T acc = T(0);
    for (int ty = ty0, iy = iy0; ty < kh; ty += up_y, ++iy) {
      if (iy < 0 || iy >= in_h) continue;
      for (int tx = tx0, ix = ix0; tx < kw; tx += up_x, ++ix) {
        if (ix < 0 || ix >= in_w) continue;
        const T kv = lds_taps ? sk[ty * kw + tx] : kernel[(kh - 1 - ty) * kw + (kw - 1 - tx)];
        acc += src[(size_t)iy * in_w + ix] * kv;
      }
    }
    out[o] = acc;
  }
}

template <typename T>
int upfirdn2d_impl(T* out, const T* in, const T* kernel, int major, int in_h, int in_w, int kh, int kw, int up_x,
                   int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream) {
  if (up_x < 1 || up_y < 1 || down_x < 1 || down_y < 1 || kh < 1 || kw < 1 || major < 0 || in_h < 0 || in_w < 0)
    return gg::fail(-2, "upfirdn2d: bad arguments");
  const int out_h = (in_h * up_y + pad_y0 + pad_y1 - kh + down_y) / down_y;
  const int out_w = (in_w * up_x + pad_x0 + pad_x1 - kw + down_x) / down_x;
  if (out_h <= 0 || out_w <= 0 || major == 0) return 0;
  if (!out || !in || !kernel) return gg::fail(-2, "upfirdn2d: null pointer");
  hipStream_t st = gg::as_stream(stream);
  const long long total = (long long)major * out_h * out_w;
  const bool blur4 = sizeof(T) == 4 && up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1 && kh == 4 && kw == 4 &&
                     out_h >= 24 && out_w >= 24;
  if (blur4) {
    const int tiles_x = (out_w + TILE - 1) / TILE, tiles_y = (out_h + TILE - 1) / TILE;
    const long long ntiles = (long long)tiles_x * tiles_y * major;
    if (ntiles < (1LL << 31)) {
      upfirdn2d_blur4_tile<false, false><<<(unsigned)ntiles, 256, 0, st>>>(
          reinterpret_cast<float*>(out), reinterpret_cast<const float*>(in), reinterpret_cast<const float*>(kernel),
          major, in_h, in_w, out_h, out_w, pad_x0, pad_y0, tiles_x, tiles_y, (unsigned)ntiles, BlurFuse{});
      return gg::launch_status("upfirdn2d_blur4_tile");
    }
  }
  upfirdn2d_direct<T><<<gg::stream_grid(total, 256), 256, 0, st>>>(out, in, kernel, total, in_h, in_w, out_h, out_w,
                                                                  kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0);
  return gg::launch_status("upfirdn2d_direct");
}

}  // namespace

extern "C" int gg_upfirdn2d_f32(float* out, const float* in, const float* kernel, int major, int in_h, int in_w,
                                int kernel_h, int kernel_w, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                                int pad_x1, int pad_y0, int pad_y1, void* stream) {
  return upfirdn2d_impl<float>(out, in, kernel, major, in_h, in_w, kernel_h, kernel_w, up_x, up_y, down_x, down_y,
                               pad_x0, pad_x1, pad_y0, pad_y1, stream);
}
extern "C" int gg_blur4_fused_f32(float* out, const float* in, const float* kernel, int n, int c, int in_h, int in_w,
                                  int pad_x0, int pad_x1, int pad_y0, int pad_y1, const float* noise,
                                  const float* noise_weight, const float* act_bias, const float* ref, float alpha,
                                  float gain, void* stream) {
  const int out_h = in_h + pad_y0 + pad_y1 - 3, out_w = in_w + pad_x0 + pad_x1 - 3;
  if (n <= 0 || c <= 0 || out_h <= 0 || out_w <= 0) return 0;
  if (!out || !in || !kernel) return gg::fail(-2, "blur4_fused: null pointer");
  const bool epi = noise != nullptr, pro = ref != nullptr;
  if (epi && (!noise_weight || !act_bias)) return gg::fail(-2, "blur4_fused: noise weight / bias missing");
  if (epi && pro) return gg::fail(-2, "blur4_fused: epilogue and prologue are exclusive");
  const int tiles_x = (out_w + TILE - 1) / TILE, tiles_y = (out_h + TILE - 1) / TILE;
  const long long ntiles = (long long)tiles_x * tiles_y * n * c;
  if (ntiles >= (1LL << 31)) return gg::fail(-2, "blur4_fused: too many tiles");
  BlurFuse f;
  f.noise = noise; f.noise_w = noise_weight; f.bias = act_bias; f.ref = ref; f.alpha = alpha; f.gain = gain;
  f.channels = c;
  hipStream_t st = gg::as_stream(stream);
  if (epi)
    upfirdn2d_blur4_tile<true, false><<<(unsigned)ntiles, 256, 0, st>>>(out, in, kernel, n * c, in_h, in_w, out_h,
                                                                         out_w, pad_x0, pad_y0, tiles_x, tiles_y,
                                                                         (unsigned)ntiles, f);
  else if (pro)
    upfirdn2d_blur4_tile<false, true><<<(unsigned)ntiles, 256, 0, st>>>(out, in, kernel, n * c, in_h, in_w, out_h,
                                                                         out_w, pad_x0, pad_y0, tiles_x, tiles_y,
                                                                         (unsigned)ntiles, f);
  else
    upfirdn2d_blur4_tile<false, false><<<(unsigned)ntiles, 256, 0, st>>>(out, in, kernel, n * c, in_h, in_w, out_h,
                                                                          out_w, pad_x0, pad_y0, tiles_x, tiles_y,
                                                                          (unsigned)ntiles, f);
  return gg::launch_status("blur4_fused");
}
extern "C" int gg_upfirdn2d_f64(double* out, const double* in, const double* kernel, int major, int in_h, int in_w,
                                int kernel_h, int kernel_w, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                                int pad_x1, int pad_y0, int pad_y1, void* stream) {
  return upfirdn2d_impl<double>(out, in, kernel, major, in_h, in_w, kernel_h, kernel_w, up_x, up_y, down_x, down_y,
                                pad_x0, pad_x1, pad_y0, pad_y1, stream);
}
