// a6  anti-aliased grid sampling (MipmapWarp / Warp of antialiased_sampling.py) as fused kernels.
//
// The reference materialises a Gaussian *stack* (N, C*D, H, W): D-1 blur+decimate steps, each
// bilinearly upsampled back to full resolution, then F.grid_sample over C*D channels, then a
// gather + lerp, with a device->host sync to learn D (antialiased_sampling.py:52).  On MI355X the
// whole thing is latency/launch bound (tiny FLOPs), so the design is:
//   1. pyramid kernels produce the *un-upsampled* levels (1/4, 1/16, 1/64 of the image);
//   2. ONE sampling kernel per pass: a thread owns an output pixel, derives its mip level from
//      the grid neighbourhood, and fetches the 4 bilinear taps of the two bracketing levels
//      directly from the pyramid - the x2^l bilinear upsample is folded into the tap fetch
//      (4 sub-taps).  No stack, no D, no host sync; identical arithmetic per tap.
// Integer by-products (floor(ix), floor(iy), floor/ceil level) follow the IEEE evaluation order
// of ATen's GridSampler.h:27-160 and antialiased_sampling.py:62-97,181-238 with contraction
// disabled (gg::mul_rn & co.), so they are bit-exact against the fp32 CPU evaluation.
#include "../../include/gangealing_hip.h"
#include "gg_common.h"

// The index / level arithmetic must be the plain IEEE sequence (mul, add, div each rounded once):
// hipcc's default -ffp-contract=fast would fuse e.g. dx*dx + dy*dy into an FMA, which changes the
// last bit, and with it floor() results and the arg-max tie-breaking of the level selection.
// (__fmul_rn & co. compile to ordinary operators unless OCML_BASIC_ROUNDED_OPERATIONS is defined,
// so they do not prevent contraction by themselves.)
#pragma clang fp contract(off)

namespace {

using gg::add_rn;
using gg::div_rn;
using gg::mul_rn;
using gg::sub_rn;

enum { PAD_ZEROS = 0, PAD_BORDER = 1, PAD_REFLECTION = 2 };

// ---------------------------------------------------------------- pyramid step (fwd / adjoint)

__device__ __forceinline__ int reflect1(int i, int n) {     // ReflectionPad2d(1) index map
  if (i < 0) return -i;
  if (i >= n) return 2 * n - 2 - i;
  return i;
}

__global__ __launch_bounds__(256) void mip_down2x_kernel(float* __restrict__ out, const float* __restrict__ in,
                                                         long long total, int h, int w) {
  const int oh = h >> 1, ow = w >> 1;
  const float f[4] = {1.f / 8.f, 3.f / 8.f, 3.f / 8.f, 1.f / 8.f};   // outer product = [1,3,3,1]^2/64 exactly
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const int ox = (int)(o % ow);
    const long long q = o / ow;
    const int oy = (int)(q % oh);
    const float* src = in + (size_t)(q / oh) * h * w;
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const int iy = reflect1(2 * oy + ky - 1, h);
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int ix = reflect1(2 * ox + kx - 1, w);
        acc += src[(size_t)iy * w + ix] * (f[ky] * f[kx]);
      }
    }
    out[o] = acc;
  }
}

__global__ __launch_bounds__(256) void mip_down2x_bwd_kernel(float* __restrict__ gin, const float* __restrict__ gout,
                                                             long long total, int h, int w) {
  const int oh = h >> 1, ow = w >> 1;
  const float f[4] = {1.f / 8.f, 3.f / 8.f, 3.f / 8.f, 1.f / 8.f};
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const int ox = (int)(o % ow);
    const long long q = o / ow;
    const int oy = (int)(q % oh);
    float* dst = gin + (size_t)(q / oh) * h * w;
    const float g = gout[o];
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const int iy = reflect1(2 * oy + ky - 1, h);
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int ix = reflect1(2 * ox + kx - 1, w);
        unsafeAtomicAdd(dst + (size_t)iy * w + ix, g * (f[ky] * f[kx]));
      }
    }
  }
}

// ---------------------------------------------------------------- coordinate helpers

__device__ __forceinline__ float unnormalize(float g, int size) {            // GridSampler.h:27-35
  return div_rn(sub_rn(mul_rn(add_rn(g, 1.f), (float)size), 1.f), 2.f);
}

__device__ __forceinline__ float clip_coord(float v, int size, float* grad) {  // GridSampler.h:58-86
  if (v <= 0.f) { *grad = 0.f; return 0.f; }
  const float mx = (float)(size - 1);
  if (v >= mx) { *grad = 0.f; return mx; }
  *grad = 1.f;
  return v;
}

__device__ __forceinline__ float reflect_coord(float v, int twice_low, int twice_high, float* grad) {  // :89-141
  if (twice_low == twice_high) { *grad = 0.f; return 0.f; }
  const float mn = (float)twice_low / 2.f;
  const float span = (float)(twice_high - twice_low) / 2.f;
  v = sub_rn(v, mn);
  float mult = 1.f;
  if (v < 0.f) { mult = -1.f; v = -v; }
  const float extra = fmodf(v, span);
  const int flips = (int)floorf(div_rn(v, span));
  if ((flips & 1) == 0) { *grad = mult; return add_rn(extra, mn); }
  *grad = -mult;
  return add_rn(sub_rn(span, extra), mn);
}

// Source coordinate of F.grid_sample (align_corners=False) and d(coord)/d(grid value).
__device__ __forceinline__ float source_coord(float g, int size, int padding_mode, float* dgrid) {
  float v = unnormalize(g, size);
  float mult = (float)size / 2.f;
  if (padding_mode == PAD_BORDER) {
    float gc;
    v = clip_coord(v, size, &gc);
    mult *= gc;
  } else if (padding_mode == PAD_REFLECTION) {
    float gr, gc;
    v = reflect_coord(v, -1, 2 * size - 1, &gr);
    v = clip_coord(v, size, &gc);
    mult *= gr * gc;
  }
  *dgrid = mult;
  return v;
}

// Absolute coordinates used for level selection: (size-1)*(g+1)/2  (antialiased_sampling.py:192-193).
__device__ __forceinline__ float level_coord(float g, int size) {
  return div_rn(mul_rn((float)(size - 1), add_rn(g, 1.f)), 2.f);
}

struct LevelInfo {
  float level;      // clamped fractional level
  int lo, hi;       // floor / ceil
  float frac;       // level % 1
  // gradient routing of d(level)/d(coords): neighbour index (0 l,1 r,2 u,3 d), -1 if blocked
  int arg;
  float dcoef;      // d level / d(sq_dist of the arg neighbour), 0 if blocked
  float ddx, ddy;   // (other - self) of the arg neighbour
};

// pin: -1, or the neighbour (0 l, 1 r, 2 u, 3 d) that the level's sub-gradient is routed through instead of this
// evaluation's own arg-max (decision replay, tests/test_gpu_stn_decisions.py: under a similarity warp the four distances
// are exactly tied in real arithmetic, so which one "is" the maximum is last-ulp noise in every implementation).  The
// level itself - the forward value - is the true maximum's in either case.
__device__ __forceinline__ LevelInfo mip_level(const float* __restrict__ grid_n, int oy, int ox, int ho, int wo,
                                              int h, int w, float max_level, float min_level, int pin = -1) {
  const float* g = grid_n + ((size_t)oy * wo + ox) * 2;
  const float cx = level_coord(g[0], w), cy = level_coord(g[1], h);
  const int nx[4] = {max(ox - 1, 0), min(ox + 1, wo - 1), ox, ox};        // replicate pad (:73-80)
  const int ny[4] = {oy, oy, max(oy - 1, 0), min(oy + 1, ho - 1)};
  float dmax = 0.f, sqmax = 1.f, bdx = 0.f, bdy = 0.f;
  int arg = -1;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float* q = grid_n + ((size_t)ny[k] * wo + nx[k]) * 2;
    const float dx = sub_rn(level_coord(q[0], w), cx), dy = sub_rn(level_coord(q[1], h), cy);
    const float sq = add_rn(mul_rn(dx, dx), mul_rn(dy, dy));
    const float sqc = fmaxf(sq, 1.f);                                     // clamp BEFORE the sqrt (:84-87)
    const float d = sqrtf(sqc);
    if (!(d <= dmax)) { dmax = d; arg = k; sqmax = sq; bdx = dx; bdy = dy; }   // first max wins
  }
  float dgrad = dmax;                    // the distance the gradient flows through
  if (pin >= 0 && pin < 4 && pin != arg) {
    const float* q = grid_n + ((size_t)ny[pin] * wo + nx[pin]) * 2;
    bdx = sub_rn(level_coord(q[0], w), cx);
    bdy = sub_rn(level_coord(q[1], h), cy);
    sqmax = add_rn(mul_rn(bdx, bdx), mul_rn(bdy, bdy));
    dgrad = sqrtf(fmaxf(sqmax, 1.f));
    arg = pin;
  }
  LevelInfo li;
  float lv = log2f(dmax);
  // integer part from the exponent of dmax (>= 1): independent of log2f's last-ulp rounding
  int e;
  const float m = frexpf(dmax, &e);                 // dmax = m * 2^e, m in [0.5, 1)
  int lo = e - 1;
  int hi = (m == 0.5f) ? lo : lo + 1;
  bool blocked = false;
  if (lv >= max_level) {                            // clamp(max = max_num_levels - 1) (:208-209)
    blocked = lv > max_level;
    lv = max_level;
    lo = (int)floorf(max_level);
    hi = (int)ceilf(max_level);
  }
  if (lv <= min_level) {                            // levels.clamp(min=min_level) (:49); also the 0 floor
    blocked = blocked || lv < min_level;
    lv = min_level;
    lo = (int)floorf(min_level);
    hi = (int)ceilf(min_level);
  }
  li.level = lv;
  li.lo = lo;
  li.hi = hi;
  li.frac = fmaxf(lv - (float)lo, 0.f);
  li.arg = arg;
  // d level / d sq = 1/(ln2 * dmax) * 0.5/sqrt(sq)  when sq >= 1 (clamp passes gradient inclusively)
  li.dcoef = (!blocked && sqmax >= 1.f) ? (1.4426950408889634f / dgrad) * (0.5f / dgrad) : 0.f;
  li.ddx = bdx;
  li.ddy = bdy;
  return li;
}

struct Axis { int i0, i1; float l0, l1; };

// Sub-taps of stack level `lvl` at full-resolution (padded) index p: the x2^lvl bilinear upsample of
// the pyramid level (F.interpolate, align_corners=False: src = (p+0.5)/2^lvl - 0.5 clamped at 0).
__device__ __forceinline__ Axis level_axis(int p, int lvl, int size_l) {
  Axis a;
  if (lvl == 0) { a.i0 = p; a.i1 = p; a.l0 = 1.f; a.l1 = 0.f; return a; }
  const float rscale = 1.f / (float)(1 << lvl);
  float src = sub_rn(mul_rn(rscale, add_rn((float)p, 0.5f)), 0.5f);
  if (src < 0.f) src = 0.f;
  a.i0 = min((int)src, size_l - 1);
  a.i1 = a.i0 + (a.i0 < size_l - 1 ? 1 : 0);
  a.l1 = src - (float)a.i0;
  a.l0 = 1.f - a.l1;
  return a;
}

struct Taps {
  int x0, y0;                 // floor(ix), floor(iy)   (bit-exact outputs)
  float wx0, wx1, wy0, wy1;
  bool vx0, vx1, vy0, vy1;    // in-bounds flags
  float dgx, dgy;             // d ix / d gx, d iy / d gy
  float ix, iy;
};

__device__ __forceinline__ Taps make_taps(const float* g, int h, int w, int padding_mode) {
  Taps t;
  t.ix = source_coord(g[0], w, padding_mode, &t.dgx);
  t.iy = source_coord(g[1], h, padding_mode, &t.dgy);
  const float fx = floorf(t.ix), fy = floorf(t.iy);
  t.x0 = (int)fx;
  t.y0 = (int)fy;
  t.wx1 = t.ix - fx;
  t.wx0 = 1.f - t.wx1;
  t.wy1 = t.iy - fy;
  t.wy0 = 1.f - t.wy1;
  t.vx0 = t.x0 >= 0 && t.x0 < w;
  t.vx1 = t.x0 + 1 >= 0 && t.x0 + 1 < w;
  t.vy0 = t.y0 >= 0 && t.y0 < h;
  t.vy1 = t.y0 + 1 >= 0 && t.y0 + 1 < h;
  return t;
}

// Un-upsampled pyramid: level 0 = the (padded) image, levels 1 .. nlev-1 consecutively in ONE buffer (`rest`), level l
// as (planes, hp >> l, wp >> l).  MAXL = how many level pointers a kernel instantiation carries: 4 (the heads'
// max_num_levels = 3.5, warping_heads.py:32,170 - the training path) or 8 (the reference's default constructor,
// antialiased_sampling.py:22: levels up to 7).  A lane picks its two levels from the array with a select chain, so the
// 4-level instantiation keeps the training path at its round-1..5 cost.
constexpr int kMaxLevels = 8;
template <int MAXL> struct PyrT { const float* p[MAXL]; };
template <int MAXL> struct GPyrT { float* p[MAXL]; };

template <int MAXL, typename P, typename T>
P make_pyr(T* base, T* rest, int nlev, long long planes, int hp, int wp) {
  P r;
  long long off = 0;
  for (int l = 0; l < MAXL; ++l) {
    if (l == 0) { r.p[0] = base; continue; }
    if (l < nlev && rest) {
      r.p[l] = rest + off;
      off += planes * (long long)(hp >> l) * (wp >> l);
    } else {
      r.p[l] = r.p[l - 1];          // never selected: levels are clamped to nlev - 1
    }
  }
  return r;
}

// Values of the 4 bilinear taps (nw, ne, sw, se) of stack level `lvl`, channel plane `plane`.
__device__ __forceinline__ void tap_values(const float* __restrict__ img, int wl, const Axis ay[2], const Axis ax[2],
                                           const Taps& t, float v[4]) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool valid = (j ? t.vy1 : t.vy0) && (i ? t.vx1 : t.vx0);
      float r = 0.f;
      if (valid) {
        const float* r0 = img + (size_t)ay[j].i0 * wl;
        const float* r1 = img + (size_t)ay[j].i1 * wl;
        const float top = ax[i].l0 * r0[ax[i].i0] + ax[i].l1 * r0[ax[i].i1];
        const float bot = ax[i].l0 * r1[ax[i].i0] + ax[i].l1 * r1[ax[i].i1];
        r = ay[j].l0 * top + ay[j].l1 * bot;
      }
      v[j * 2 + i] = r;
    }
  }
}

__device__ __forceinline__ void make_axes(const Taps& t, int lvl, int h, int w, int hp, int wp, int pad_l,
                                          Axis ay[2], Axis ax[2]) {
  const int hl = hp >> lvl, wl = wp >> lvl;
  // clamp only guards the address computation of masked-out taps
  ay[0] = level_axis(min(max(t.y0, 0), h - 1) + pad_l, lvl, hl);
  ay[1] = level_axis(min(max(t.y0 + 1, 0), h - 1) + pad_l, lvl, hl);
  ax[0] = level_axis(min(max(t.x0, 0), w - 1) + pad_l, lvl, wl);
  ax[1] = level_axis(min(max(t.x0 + 1, 0), w - 1) + pad_l, lvl, wl);
}

// ---------------------------------------------------------------- forward

template <int MAXL>
__global__ __launch_bounds__(256) void mipmap_warp_fwd_kernel(
    float* __restrict__ out, float* __restrict__ levels_out, PyrT<MAXL> pyr, const float* __restrict__ grid, int n, int c,
    int h, int w, int hp, int wp, int pad_l, int ho, int wo, float max_level, float min_level, int padding_mode,
    int antialias, int top) {
  const long long total = (long long)n * ho * wo;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const int ox = (int)(o % wo);
    const long long q = o / wo;
    const int oy = (int)(q % ho);
    const int s = (int)(q / ho);
    const float* grid_n = grid + (size_t)s * ho * wo * 2;
    LevelInfo li;
    if (antialias) {
      li = mip_level(grid_n, oy, ox, ho, wo, h, w, max_level, min_level);
    } else {
      li.level = 0.f; li.lo = 0; li.hi = 0; li.frac = 0.f;
    }
    if (levels_out) levels_out[o] = li.level;
    // `top` = the deepest level the pyramid holds (its 1 x 1 level at the latest): a deeper request - where the
    // reference's ReflectionPad2d(1) of a 1 x 1 map raises (antialiased_sampling.py:117) - samples that level
    li.lo = min(li.lo, top);
    li.hi = min(li.hi, top);
    const Taps t = make_taps(grid_n + ((size_t)oy * wo + ox) * 2, h, w, padding_mode);
    const float wt[4] = {t.wy0 * t.wx0, t.wy0 * t.wx1, t.wy1 * t.wx0, t.wy1 * t.wx1};
    Axis ay0[2], ax0[2], ay1[2], ax1[2];
    make_axes(t, li.lo, h, w, hp, wp, pad_l, ay0, ax0);
    const bool two = li.hi != li.lo;
    if (two) make_axes(t, li.hi, h, w, hp, wp, pad_l, ay1, ax1);
    const int wl0 = wp >> li.lo, hl0 = hp >> li.lo, wl1 = wp >> li.hi, hl1 = hp >> li.hi;
    for (int ch = 0; ch < c; ++ch) {
      const size_t plane = (size_t)s * c + ch;
      float v[4];
      tap_values(pyr.p[li.lo] + plane * hl0 * wl0, wl0, ay0, ax0, t, v);
      const float o0 = v[0] * wt[0] + v[1] * wt[1] + v[2] * wt[2] + v[3] * wt[3];
      float res = o0;
      if (two) {
        tap_values(pyr.p[li.hi] + plane * hl1 * wl1, wl1, ay1, ax1, t, v);
        const float o1 = v[0] * wt[0] + v[1] * wt[1] + v[2] * wt[2] + v[3] * wt[3];
        res = o0 + li.frac * (o1 - o0);
      }
      out[(plane * ho + oy) * wo + ox] = res;
    }
  }
}

// Integer by-products of the sampling, produced by the SAME device functions the forward / backward kernels call
// (make_taps, mip_level): floor(ix), floor(iy) after unnormalise + padding-mode coordinate transform
// (GridSampler.h:143-160 + floor in grid_sampler_2d), floor / ceil of the clamped mip level
// (antialiased_sampling.py:226-227).
__global__ __launch_bounds__(256) void mipmap_warp_indices_kernel(int* __restrict__ ix_nw, int* __restrict__ iy_nw,
                                                                  int* __restrict__ lvl_floor,
                                                                  int* __restrict__ lvl_ceil,
                                                                  int* __restrict__ arg_out,
                                                                  const float* __restrict__ grid, int n, int h, int w,
                                                                  int ho, int wo, float max_level, float min_level,
                                                                  int padding_mode, int antialias) {
  const long long total = (long long)n * ho * wo;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const int ox = (int)(o % wo);
    const long long q = o / wo;
    const int oy = (int)(q % ho);
    const int s = (int)(q / ho);
    const float* grid_n = grid + (size_t)s * ho * wo * 2;
    int lo = 0, hi = 0, arg = -1;
    if (antialias) {
      const LevelInfo li = mip_level(grid_n, oy, ox, ho, wo, h, w, max_level, min_level);
      lo = li.lo;
      hi = li.hi;
      arg = li.arg;
    }
    const Taps t = make_taps(grid_n + ((size_t)oy * wo + ox) * 2, h, w, padding_mode);
    if (ix_nw) ix_nw[o] = t.x0;
    if (iy_nw) iy_nw[o] = t.y0;
    if (lvl_floor) lvl_floor[o] = lo;
    if (lvl_ceil) lvl_ceil[o] = hi;
    if (arg_out) arg_out[o] = arg;
  }
}

// ---------------------------------------------------------------- backward

__device__ __forceinline__ void scatter_taps(float* __restrict__ gimg, int wl, const Axis ay[2], const Axis ax[2],
                                             const Taps& t, const float wt[4], float g) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool valid = (j ? t.vy1 : t.vy0) && (i ? t.vx1 : t.vx0);
      if (!valid) continue;
      const float gv = g * wt[j * 2 + i];
      float* r0 = gimg + (size_t)ay[j].i0 * wl;
      float* r1 = gimg + (size_t)ay[j].i1 * wl;
      unsafeAtomicAdd(r0 + ax[i].i0, gv * ay[j].l0 * ax[i].l0);
      unsafeAtomicAdd(r0 + ax[i].i1, gv * ay[j].l0 * ax[i].l1);
      unsafeAtomicAdd(r1 + ax[i].i0, gv * ay[j].l1 * ax[i].l0);
      unsafeAtomicAdd(r1 + ax[i].i1, gv * ay[j].l1 * ax[i].l1);
    }
  }
}

template <int MAXL>
__global__ __launch_bounds__(256) void mipmap_warp_bwd_kernel(
    float* __restrict__ ggrid, GPyrT<MAXL> gpyr, const float* __restrict__ gout, PyrT<MAXL> pyr,
    const float* __restrict__ grid, int n, int c, int h, int w, int hp, int wp, int pad_l, int ho, int wo,
    float max_level, float min_level, int padding_mode, int antialias, int want_image_grad,
    int* __restrict__ nb_target, float2* __restrict__ nb_grad, int top, const signed char* __restrict__ arg_pin) {
  const long long total = (long long)n * ho * wo;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const int ox = (int)(o % wo);
    const long long q = o / wo;
    const int oy = (int)(q % ho);
    const int s = (int)(q / ho);
    const float* grid_n = grid + (size_t)s * ho * wo * 2;
    float* ggrid_n = ggrid + (size_t)s * ho * wo * 2;
    LevelInfo li;
    if (antialias) {
      li = mip_level(grid_n, oy, ox, ho, wo, h, w, max_level, min_level, arg_pin ? (int)arg_pin[o] : -1);
    } else {
      li.level = 0.f; li.lo = 0; li.hi = 0; li.frac = 0.f; li.arg = -1; li.dcoef = 0.f; li.ddx = li.ddy = 0.f;
    }
    li.lo = min(li.lo, top);
    li.hi = min(li.hi, top);
    const Taps t = make_taps(grid_n + ((size_t)oy * wo + ox) * 2, h, w, padding_mode);
    const float wt[4] = {t.wy0 * t.wx0, t.wy0 * t.wx1, t.wy1 * t.wx0, t.wy1 * t.wx1};
    Axis ay0[2], ax0[2], ay1[2], ax1[2];
    make_axes(t, li.lo, h, w, hp, wp, pad_l, ay0, ax0);
    const bool two = li.hi != li.lo;
    if (two) make_axes(t, li.hi, h, w, hp, wp, pad_l, ay1, ax1);
    const int wl0 = wp >> li.lo, hl0 = hp >> li.lo, wl1 = wp >> li.hi, hl1 = hp >> li.hi;
    float gix = 0.f, giy = 0.f, gfrac = 0.f;
    for (int ch = 0; ch < c; ++ch) {
      const size_t plane = (size_t)s * c + ch;
      const float g = gout[(plane * ho + oy) * wo + ox];
      float v0[4], v1[4];
      tap_values(pyr.p[li.lo] + plane * hl0 * wl0, wl0, ay0, ax0, t, v0);
      float g0 = g, g1 = 0.f;
      if (two) {
        tap_values(pyr.p[li.hi] + plane * hl1 * wl1, wl1, ay1, ax1, t, v1);
        const float o0 = v0[0] * wt[0] + v0[1] * wt[1] + v0[2] * wt[2] + v0[3] * wt[3];
        const float o1 = v1[0] * wt[0] + v1[1] * wt[1] + v1[2] * wt[2] + v1[3] * wt[3];
        gfrac += g * (o1 - o0);
        g0 = g * (1.f - li.frac);
        g1 = g * li.frac;
      }
      // d/d(ix,iy) of the bilinear blend (GridSampler backward): taps nw, ne, sw, se
      gix += g0 * ((v0[1] - v0[0]) * t.wy0 + (v0[3] - v0[2]) * t.wy1);
      giy += g0 * ((v0[2] - v0[0]) * t.wx0 + (v0[3] - v0[1]) * t.wx1);
      if (two) {
        gix += g1 * ((v1[1] - v1[0]) * t.wy0 + (v1[3] - v1[2]) * t.wy1);
        giy += g1 * ((v1[2] - v1[0]) * t.wx0 + (v1[3] - v1[1]) * t.wx1);
      }
      if (want_image_grad) {
        scatter_taps(gpyr.p[li.lo] + plane * hl0 * wl0, wl0, ay0, ax0, t, wt, g0);
        if (two) scatter_taps(gpyr.p[li.hi] + plane * hl1 * wl1, wl1, ay1, ax1, t, wt, g1);
      }
    }
    float gx = gix * t.dgx, gy = giy * t.dgy;
    // gradient through the fractional level: level -> dist_max -> coords of self and of the arg neighbour.  The
    // neighbour's share is not added here (that would be a float atomic: up to five contributions per grid point in
    // arrival order) but recorded per SOURCE point; mipmap_warp_bwd_gather_kernel adds the records in a fixed order.
    int target = -1;
    float2 ng = make_float2(0.f, 0.f);
    if (antialias && li.arg >= 0 && li.dcoef != 0.f && gfrac != 0.f) {
      const float gsq = gfrac * li.dcoef;                    // d loss / d sq_dist
      const float gox = gsq * 2.f * li.ddx * ((float)(w - 1) / 2.f);
      const float goy = gsq * 2.f * li.ddy * ((float)(h - 1) / 2.f);
      const int nx = (li.arg == 0) ? max(ox - 1, 0) : (li.arg == 1) ? min(ox + 1, wo - 1) : ox;
      const int ny = (li.arg == 2) ? max(oy - 1, 0) : (li.arg == 3) ? min(oy + 1, ho - 1) : oy;
      target = ny * wo + nx;
      ng = make_float2(gox, goy);
      gx -= gox;
      gy -= goy;
    }
    float* gs = ggrid_n + ((size_t)oy * wo + ox) * 2;
    gs[0] = gx;
    gs[1] = gy;
    if (nb_target) {
      nb_target[o] = target;
      nb_grad[o] = ng;
    }
  }
}

// grad_grid[p] += the recorded neighbour shares aimed at p, visited in a fixed order (self - at a clamped border the
// "neighbour" is the point itself -, left, right, up, down): bitwise reproducible, no atomics
__global__ __launch_bounds__(256) void mipmap_warp_bwd_gather_kernel(float* __restrict__ ggrid,
                                                                     const int* __restrict__ nb_target,
                                                                     const float2* __restrict__ nb_grad, long long total,
                                                                     int ho, int wo) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const int ox = (int)(o % wo);
    const long long q = o / wo;
    const int oy = (int)(q % ho);
    const long long base = (q / ho) * ho * wo;
    const int self = oy * wo + ox;
    float gx = ggrid[o * 2], gy = ggrid[o * 2 + 1];
    const int cy[5] = {oy, oy, oy, oy - 1, oy + 1}, cx[5] = {ox, ox - 1, ox + 1, ox, ox};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      if ((unsigned)cy[k] >= (unsigned)ho || (unsigned)cx[k] >= (unsigned)wo) continue;
      const long long src = base + (long long)cy[k] * wo + cx[k];
      if (nb_target[src] == self) {
        const float2 g = nb_grad[src];
        gx += g.x;
        gy += g.y;
      }
    }
    ggrid[o * 2] = gx;
    ggrid[o * 2 + 1] = gy;
  }
}

int check_common(const float* grid, int n, int c, int h, int w, int hp, int wp, int pad_l, int ho, int wo,
                 int padding_mode, int antialias) {
  if (n < 0 || c < 0 || h <= 0 || w <= 0 || ho < 0 || wo < 0 || !grid) return gg::fail(-2, "mipmap_warp: bad sizes");
  if (padding_mode < 0 || padding_mode > 2) return gg::fail(-2, "mipmap_warp: padding_mode must be 0, 1 or 2");
  if (!antialias) {
    if (hp != h || wp != w || pad_l != 0) return gg::fail(-2, "warp: plain sampling takes the unpadded image");
    return 0;
  }
  if (hp != wp || (hp & (hp - 1)) != 0 || hp < h + pad_l || wp < w + pad_l || pad_l < 0)
    return gg::fail(-2, "mipmap_warp: pyramid base must be a square power of two covering the padded input");
  if (hp < 2) return gg::fail(-2, "mipmap_warp: pyramid base must be at least 2x2");
  return 0;
}

}  // namespace

extern "C" int gg_mip_downsample2x_f32(float* out, const float* in, int planes, int h, int w, void* stream) {
  if (planes <= 0) return 0;
  if (!out || !in || h < 2 || w < 2 || (h & 1) || (w & 1)) return gg::fail(-2, "mip_downsample2x: bad arguments");
  const long long total = (long long)planes * (h / 2) * (w / 2);
  mip_down2x_kernel<<<gg::stream_grid(total, 256), 256, 0, gg::as_stream(stream)>>>(out, in, total, h, w);
  return gg::launch_status("mip_downsample2x");
}

extern "C" int gg_mip_downsample2x_bwd_f32(float* grad_in, const float* grad_out, int planes, int h, int w,
                                           void* stream) {
  if (planes <= 0) return 0;
  if (!grad_in || !grad_out || h < 2 || w < 2 || (h & 1) || (w & 1))
    return gg::fail(-2, "mip_downsample2x_bwd: bad arguments");
  const long long total = (long long)planes * (h / 2) * (w / 2);
  mip_down2x_bwd_kernel<<<gg::stream_grid(total, 256), 256, 0, gg::as_stream(stream)>>>(grad_in, grad_out, total, h,
                                                                                        w);
  return gg::launch_status("mip_downsample2x_bwd");
}

// num_levels: how many pyramid levels the caller built (level 0 = pyr0, 1 .. num_levels-1 in pyr_rest); the sampling
// needs ceil(max_level) + 1 of them unless the pyramid ends earlier at its 1 x 1 level.
static int check_levels(const char* who, int antialias, float max_level, int num_levels, const void* rest, int hp,
                        int* top) {
  *top = 0;
  if (!antialias) return 0;
  if (!(max_level >= 0.f) || max_level > (float)(kMaxLevels - 1))
    return gg::fail(-2, "%s: max_level must be in [0, %d] (max_num_levels <= %d)", who, kMaxLevels - 1, kMaxLevels);
  int avail = 1;
  while ((hp >> avail) >= 1 && avail < kMaxLevels) ++avail;          // levels down to 1 x 1
  int need = (int)ceilf(max_level) + 1;
  if (need > avail) need = avail;
  if (num_levels < need || num_levels > kMaxLevels || (num_levels > 1 && !rest))
    return gg::fail(-2, "%s: antialias with max_level %g on a %d-pixel base needs %d pyramid levels, got %d", who,
                    (double)max_level, hp, need, num_levels);
  *top = num_levels - 1;
  return 0;
}

extern "C" int gg_mipmap_warp_fwd_f32(float* out, float* levels_out, const float* pyr0, const float* pyr_rest,
                                      int num_levels, const float* grid, int n, int c, int h, int w, int hp, int wp,
                                      int pad_l, int ho, int wo, float max_level, float min_level, int padding_mode,
                                      int antialias, void* stream) {
  int rc = check_common(grid, n, c, h, w, hp, wp, pad_l, ho, wo, padding_mode, antialias);
  if (rc) return rc;
  const long long total = (long long)n * ho * wo;
  if (total == 0 || c == 0) return 0;
  if (!out || !pyr0) return gg::fail(-2, "mipmap_warp_fwd: null pointer");
  int top;
  if ((rc = check_levels("mipmap_warp_fwd", antialias, max_level, num_levels, pyr_rest, hp, &top))) return rc;
  const long long planes = (long long)n * c;
  hipStream_t st = gg::as_stream(stream);
  if (top < 4) {
    auto pyr = make_pyr<4, PyrT<4>>(pyr0, pyr_rest, top + 1, planes, hp, wp);
    mipmap_warp_fwd_kernel<4><<<gg::stream_grid(total, 256), 256, 0, st>>>(
        out, levels_out, pyr, grid, n, c, h, w, hp, wp, pad_l, ho, wo, max_level, min_level, padding_mode, antialias, top);
  } else {
    auto pyr = make_pyr<8, PyrT<8>>(pyr0, pyr_rest, top + 1, planes, hp, wp);
    mipmap_warp_fwd_kernel<8><<<gg::stream_grid(total, 256), 256, 0, st>>>(
        out, levels_out, pyr, grid, n, c, h, w, hp, wp, pad_l, ho, wo, max_level, min_level, padding_mode, antialias, top);
  }
  return gg::launch_status("mipmap_warp_fwd");
}

extern "C" int gg_mipmap_warp_indices_f32(int* ix_nw, int* iy_nw, int* lvl_floor, int* lvl_ceil, int* level_arg,
                                          const float* grid, int n, int h, int w, int ho, int wo, float max_level,
                                          float min_level, int padding_mode, int antialias, void* stream) {
  if (n < 0 || h <= 0 || w <= 0 || ho < 0 || wo < 0 || !grid) return gg::fail(-2, "mipmap_warp_indices: bad sizes");
  if (padding_mode < 0 || padding_mode > 2) return gg::fail(-2, "mipmap_warp_indices: padding_mode must be 0, 1 or 2");
  if (antialias && (!(max_level >= 0.f) || max_level > (float)(kMaxLevels - 1)))
    return gg::fail(-2, "mipmap_warp_indices: max_level must be in [0, 7]");
  const long long total = (long long)n * ho * wo;
  if (total == 0) return 0;
  mipmap_warp_indices_kernel<<<gg::stream_grid(total, 256), 256, 0, gg::as_stream(stream)>>>(
      ix_nw, iy_nw, lvl_floor, lvl_ceil, level_arg, grid, n, h, w, ho, wo, max_level, min_level, padding_mode, antialias);
  return gg::launch_status("mipmap_warp_indices");
}

extern "C" int gg_mipmap_warp_bwd_f32(float* grad_grid, float* grad_pyr0, float* grad_pyr_rest, const float* grad_out,
                                      const float* pyr0, const float* pyr_rest, int num_levels, const float* grid,
                                      int n, int c, int h, int w, int hp, int wp, int pad_l, int ho, int wo,
                                      float max_level, float min_level, int padding_mode, int antialias,
                                      const signed char* level_arg_pin, void* stream) {
  int rc = check_common(grid, n, c, h, w, hp, wp, pad_l, ho, wo, padding_mode, antialias);
  if (rc) return rc;
  const long long total = (long long)n * ho * wo;
  if (total == 0) return 0;
  if (!grad_grid || !grad_out || !pyr0) return gg::fail(-2, "mipmap_warp_bwd: null pointer");
  int top;
  if ((rc = check_levels("mipmap_warp_bwd", antialias, max_level, num_levels, pyr_rest, hp, &top))) return rc;
  const int want_img = grad_pyr0 != nullptr;
  if (want_img && antialias && top > 0 && !grad_pyr_rest)
    return gg::fail(-2, "mipmap_warp_bwd: the image gradient needs grad_pyr_rest (same layout as pyr_rest)");
  hipStream_t st = gg::as_stream(stream);
  if (c == 0) {
    hipError_t e = hipMemsetAsync(grad_grid, 0, sizeof(float) * (size_t)total * 2, st);
    return e == hipSuccess ? 0 : gg::fail((int)e, "mipmap_warp_bwd: memset failed");
  }
  if ((long long)ho * wo >= (1LL << 31)) return gg::fail(-2, "mipmap_warp_bwd: output too large");
  // neighbour records (one int + one float2 per output point) live in the stream's scratch
  int* nb_target = nullptr;
  float2* nb_grad = nullptr;
  if (antialias) {
    char* sc = reinterpret_cast<char*>(gg::scratch(st, (size_t)total * 12));
    if (!sc) return -3;
    nb_grad = reinterpret_cast<float2*>(sc);
    nb_target = reinterpret_cast<int*>(sc + (size_t)total * 8);
  }
  const long long planes = (long long)n * c;
  if (top < 4) {
    auto pyr = make_pyr<4, PyrT<4>>(pyr0, pyr_rest, top + 1, planes, hp, wp);
    auto gp = make_pyr<4, GPyrT<4>>(grad_pyr0, grad_pyr_rest, top + 1, planes, hp, wp);
    mipmap_warp_bwd_kernel<4><<<gg::stream_grid(total, 256), 256, 0, st>>>(
        grad_grid, gp, grad_out, pyr, grid, n, c, h, w, hp, wp, pad_l, ho, wo, max_level, min_level, padding_mode,
        antialias, want_img, nb_target, nb_grad, top, level_arg_pin);
  } else {
    auto pyr = make_pyr<8, PyrT<8>>(pyr0, pyr_rest, top + 1, planes, hp, wp);
    auto gp = make_pyr<8, GPyrT<8>>(grad_pyr0, grad_pyr_rest, top + 1, planes, hp, wp);
    mipmap_warp_bwd_kernel<8><<<gg::stream_grid(total, 256), 256, 0, st>>>(
        grad_grid, gp, grad_out, pyr, grid, n, c, h, w, hp, wp, pad_l, ho, wo, max_level, min_level, padding_mode,
        antialias, want_img, nb_target, nb_grad, top, level_arg_pin);
  }
  rc = gg::launch_status("mipmap_warp_bwd");
  if (rc || !antialias) return rc;
  mipmap_warp_bwd_gather_kernel<<<gg::stream_grid(total, 256), 256, 0, st>>>(grad_grid, nb_target, nb_grad, total, ho,
                                                                             wo);
  return gg::launch_status("mipmap_warp_bwd_gather");
}
