// a3/a4  convolutions as implicit GEMM on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
// GEMM orientation (chosen for NCHW stores on 64-wide waves):
//     D[co, pix] = sum_k  Wm[k, co] * X[k, pix]        k = (ci, tap)
//   MFMA "A" operand = weights   (lane l: co = l & 31, k = l >> 5)
//   MFMA "B" operand = im2col'ed activations gathered on the fly (lane l: k = l >> 5, pix = l & 31)
//   D: lane l owns pixel (l & 31) and 16 output channels -> for every accumulator register the 32
//   lanes of a half-wave store 32 consecutive pixels of ONE channel plane = 128 B segments.
// Both LDS tiles are [k][x] with x contiguous, so every ds_read_b32 of a fragment hits 32
// consecutive banks (conflict free) and needs no swizzle.
//
// Block = 256 threads = 4 waves.  Two tilings:
//   big   : 128 co x 128 pix, waves 2x2, each wave 2x2 MFMA tiles (64 accumulator VGPRs)
//   narrow:  32 co x 256 pix, waves 1x4, each wave 1x2 MFMA tiles  (ToRGB / flow heads, Cout <= 32)
// K is consumed in BK = 16 slabs, register-staged double buffering (global loads of slab t+1 are in
// flight while slab t is multiplied; one __syncthreads per slab).  Small spatial layers (4^2..16^2)
// do not produce 256 tiles, so K is split across blockIdx.y; the splits' raw partial sums go to the stream's scratch
// and a reduce pass adds them in a fixed order and applies the epilogue (no float atomics: bitwise reproducible).
//
// The modulated convolution (networks.py:233-282) is run in its shared-weight form: the per-sample
// style multiplies the activation while it is gathered (in_scale) and the demodulation multiplies
// the accumulator in the epilogue (out_scale); no (N,Cout,Cin,3,3) weight tensor ever exists.
//
// Stride-2 transposed convolution (the generator's up-convs and the dgrad of the STN's strided
// convs) is decomposed by output parity into 4 dense sub-problems (2x2, 2x1, 1x2, 1x1 taps), so no
// MFMA work is spent on the zero-stuffed positions.
//
// Kernel families in this file (DESIGN.md section 3 has the measurements):
//   conv_igemm_kernel            exact fp32 MFMA implicit GEMM, every shape (parity mode; 3-channel stems, ToRGB)
//   conv_split_kernel            split-precision (2 / 3 bf16 limbs per fp32 operand) generic shapes: 1x1, stride 2
//   conv3x3_patch_kernel         split-precision 3x3 / stride 1 with input-patch reuse, fused StyledConv epilogue and
//                                optional leaky-ReLU mask on the input (data gradients) - the dominant kernel
//   convT3x3s2_patch_kernel      split-precision transposed 3x3 / stride 2, all four parity classes in one pass
//   conv_wgrad_kernel / conv_wgrad_split_kernel / conv3x3_wgrad_rows_kernel   weight gradients (fp32 / generic split /
//                                row-streaming 3x3 with workspace reduction)
//   pack_weight*_kernel          GEMM / limb-plane weight layouts (single, or all trainable weights in one launch)
#include "../../include/gangealing_hip.h"
#include "gg_common.h"
#include <string.h>
#include "conv_common.h"

namespace {

using namespace gg_conv;

// k -> (ci, ky, kx, dy, dx).  MODE 0: correlation taps; MODE 1: taps of one parity class.
template <int KS, int MODE>
__device__ __forceinline__ void decode_k(const ConvArgs& a, int k, int& ci, int& ky, int& kx, int& dy, int& dx) {
  if (MODE == 0) {
    constexpr int KK = KS * KS;
    ci = k / KK;
    const int tap = k - ci * KK;
    ky = tap / KS;
    kx = tap - ky * KS;
    dy = ky;
    dx = kx;
  } else {
    const int ntaps = a.nty * a.ntx;          // 1, 2 or 4 (ntx, nty in {1, 2})
    const int sh = (ntaps == 4) ? 2 : (ntaps == 2 ? 1 : 0);
    ci = k >> sh;
    const int tap = k & (ntaps - 1);
    const int jy = (a.ntx == 2) ? (tap >> 1) : tap;
    const int jx = (a.ntx == 2) ? (tap & 1) : 0;
    ky = a.py + 2 * jy;
    kx = a.px + 2 * jx;
    dy = -jy;
    dx = -jx;
  }
}

template <int KS, int MODE, int WCO, int WPIX, int MI, int NJ, bool IN_SCALE>
__global__ __launch_bounds__(WCO * WPIX * 64) void conv_igemm_kernel(const ConvArgs a) {
  constexpr int NT = WCO * WPIX * 64;           // threads per block
  constexpr int TCO = WCO * MI * 32;
  constexpr int TPIX = WPIX * NJ * 32;
  constexpr int ROWS_A = NT / TPIX;             // k rows covered per pass by the pixel gather
  constexpr int PASS_A = BK / ROWS_A;
  constexpr int ROWS_B = NT / TCO;
  constexpr int PASS_B = BK / ROWS_B;
  constexpr int KK = KS * KS;
  __shared__ float sX[2][BK][TPIX];
  __shared__ float sW[2][BK][TCO];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wco = wid / WPIX, wpix = wid % WPIX;
  const unsigned ntiles = (unsigned)a.tiles_co * a.tiles_pix;
  const unsigned logical = gg::xcd_remap(blockIdx.x, ntiles);
  const int tile_co = logical % a.tiles_co, tile_pix = logical / a.tiles_co;
  const int split = blockIdx.y, g = blockIdx.z;
  const int co0 = tile_co * TCO;
  const long long m0 = (long long)tile_pix * TPIX;
  const long long mtot = (long long)a.batch * a.mh * a.mw;
  const int hw = a.h * a.w;

  // ---- per-thread gather column (one pixel of the tile) ----
  const int pcol = tid % TPIX, prow = tid / TPIX;
  const long long m = m0 + pcol;
  const bool m_ok = m < mtot;
  int pn = 0, base_y = 0, base_x = 0;
  if (m_ok) {
    const int per = a.mh * a.mw;
    pn = (int)(m / per);
    const int rem = (int)(m - (long long)pn * per);
    const int qy = rem / a.mw, qx = rem - qy * a.mw;
    base_y = qy * a.bs + a.byo;
    base_x = qx * a.bs + a.bxo;
  }
  const int chan0 = (pn * a.groups + g) * a.cin_g;           // first input channel of this (sample, group)
  const float* xg = a.x + (size_t)chan0 * hw;
  const float* sg = IN_SCALE ? a.in_scale + chan0 : nullptr;
  // ---- per-thread weight column ----
  const int ccol = tid % TCO, crow = tid / TCO;
  const bool c_ok = (co0 + ccol) < a.cout_g;
  const float* wg_safe = a.wmat + (size_t)g * ((size_t)a.cin_g * KK) * a.cout_g + (c_ok ? co0 + ccol : 0);

  const int slab0 = split * a.slabs_per_split;
  int slab1 = slab0 + a.slabs_per_split;
  if (slab1 > a.nslabs) slab1 = a.nslabs;

  // Register staging of the next slab.  Loads are unconditional (addresses clamped to a legal
  // element, validity kept in a bit mask) so that all of a slab's global loads are issued back to
  // back and retire under the MFMAs of the current slab; masking and the style scale are applied
  // when the registers are written to LDS.
  float ra[PASS_A], rs[PASS_A], rb[PASS_B];
  unsigned mask_a = 0, mask_b = 0;

  auto load_a = [&](int kbase, int p) {
    const int k = kbase + prow + p * ROWS_A;
    int ci, ky, kx, dy, dx;
    decode_k<KS, MODE>(a, min(k, a.ktot - 1), ci, ky, kx, dy, dx);
    const int iy = base_y + dy, ix = base_x + dx;
    // branch-free validity: unsigned compares fold the >= 0 tests
    const unsigned ok = (unsigned)m_ok & (unsigned)(k < a.ktot) & (unsigned)((unsigned)iy < (unsigned)a.h) &
                        (unsigned)((unsigned)ix < (unsigned)a.w);
    const int off = (ci * hw + iy * a.w + ix) * (int)ok;
    ra[p] = xg[off];
    if (IN_SCALE) rs[p] = sg[ci];
    mask_a |= ok << p;
  };
  auto load_b = [&](int kbase, int p) {
    const int k = kbase + crow + p * ROWS_B;
    int ci, ky, kx, dy, dx;
    decode_k<KS, MODE>(a, min(k, a.ktot - 1), ci, ky, kx, dy, dx);
    const unsigned ok = (unsigned)c_ok & (unsigned)(k < a.ktot);
    rb[p] = wg_safe[(size_t)(ci * KK + ky * KS + kx) * a.cout_g];
    mask_b |= ok << p;
  };
  auto load_slab = [&](int slab) {
    const int kbase = slab * BK;
    mask_a = 0;
    mask_b = 0;
#pragma unroll
    for (int p = 0; p < PASS_A; ++p) load_a(kbase, p);
#pragma unroll
    for (int p = 0; p < PASS_B; ++p) load_b(kbase, p);
  };
  auto store_slab = [&](int buf) {
#pragma unroll
    for (int p = 0; p < PASS_A; ++p) sX[buf][prow + p * ROWS_A][pcol] = ((mask_a >> p) & 1u) ? (IN_SCALE ? ra[p] * rs[p] : ra[p]) : 0.f;
#pragma unroll
    for (int p = 0; p < PASS_B; ++p) sW[buf][crow + p * ROWS_B][ccol] = ((mask_b >> p) & 1u) ? rb[p] : 0.f;
  };

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (slab0 < slab1) {
    load_slab(slab0);
    store_slab(0);
    __syncthreads();
    int cur = 0;
    const int kh = lane >> 5, l31 = lane & 31;
    for (int slab = slab0; slab < slab1; ++slab) {
      const bool more = slab + 1 < slab1;
      // global loads of slab t+1 are issued first and retire under the 32 MFMAs of slab t (measured:
      // spreading them between the MFMAs with sched_barrier pinning was 4 % slower)
      if (more) load_slab(slab + 1);
      // fragments are double-buffered in registers: the LDS reads of k-step kk+1 are issued before the
      // MFMAs of k-step kk, so their latency hides under 4 x 64 cycles of matrix pipe
      float wa[2][MI], xb[2][NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) wa[0][i] = sW[cur][kh][(wco * MI + i) * 32 + l31];
#pragma unroll
      for (int j = 0; j < NJ; ++j) xb[0][j] = sX[cur][kh][(wpix * NJ + j) * 32 + l31];
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        const int c = kk & 1, nx = c ^ 1;
        if (kk + 1 < BK / 2) {
#pragma unroll
          for (int i = 0; i < MI; ++i) wa[nx][i] = sW[cur][(kk + 1) * 2 + kh][(wco * MI + i) * 32 + l31];
#pragma unroll
          for (int j = 0; j < NJ; ++j) xb[nx][j] = sX[cur][(kk + 1) * 2 + kh][(wpix * NJ + j) * 32 + l31];
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[c][i], xb[c][j], acc[i][j], 0, 0, 0);
        // pin the issue order: this step's LDS reads (for the NEXT k-step) first, then its MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, MI + NJ, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, MI * NJ, 0);
      }
      if (more) store_slab(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }

  // ---- epilogue: D[co][pix], lane -> pixel (lane & 31), reg r -> co = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int ohw = a.oh * a.ow;
  const bool partial = a.part != nullptr;
  float* ybase = partial ? a.part + (size_t)split * a.part_stride : a.y;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const long long mm = m0 + (wpix * NJ + j) * 32 + (lane & 31);
    if (mm >= mtot) continue;
    const int per = a.mh * a.mw;
    const int on = (int)(mm / per);
    const int rem = (int)(mm - (long long)on * per);
    const int qy = rem / a.mw, qx = rem - qy * a.mw;
    const int oy = qy * a.ys + a.yo, ox = qx * a.xs + a.xo;
    const int ochan0 = (on * a.groups + g) * a.cout_g;
    float* yp = ybase + (size_t)ochan0 * ohw + (size_t)oy * a.ow + ox;
    const float* osc = (a.out_scale && !partial) ? a.out_scale + ochan0 : nullptr;
    const float* bia = (a.bias && !partial) ? a.bias + g * a.cout_g : nullptr;
    const float* res = (a.residual && !partial) ? a.residual + (size_t)ochan0 * ohw + (size_t)oy * a.ow + ox : nullptr;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + (wco * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co >= a.cout_g) continue;
        float v = acc[i][j][r];
        if (osc) v *= osc[co];
        if (bia) v += bia[co];
        if (res) v += res[(size_t)co * ohw];
        yp[(size_t)co * ohw] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Split-precision variant: the same implicit GEMM on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16,
// 16x the fp32 MFMA rate).  Every fp32 operand is split into LIMBS bf16 limbs (x = x0 + x1 (+ x2),
// each limb the bf16 rounding of the remaining residual) and the product is assembled from the limb
// pairs (i, j) with i + j < LIMBS, accumulated in fp32:
//     LIMBS = 2 -> 3 MFMAs per tile step, |error| ~ 2^-16 |a||b| per product  ("bf16x3")
//     LIMBS = 3 -> 6 MFMAs per tile step, |error| ~ 2^-23 |a||b|: fp32-class ("bf16x6")
// K is ordered (tap, ci) with ci fastest, so a 32-wide K slab is ONE tap and 32 consecutive input
// channels: one bounds test and one base address per thread per slab, 16 loads at a constant channel
// stride (each wave-load is 64 consecutive pixels), then cvt_pk splits.  LDS tiles are [row][k] with
// k contiguous (80-byte rows: 5 x 16 B slots, odd -> conflict-free ds_read_b128 / ds_write_b128);
// a fragment is one 16-byte read.  Single LDS buffer + register prefetch, two barriers per slab.
// ------------------------------------------------------------------------------------------------
// (typedefs, buffer access, Limb<>, BlockExp: conv_common.h)

template <int KS, int MODE, int LIMBS, bool IN_SCALE, int TPIX, bool F16 = false>
__global__ __launch_bounds__(TPIX * 2, 2) void conv_split_kernel(const ConvArgs a) {
  using L = Limb<F16>;
  constexpr int TCO = 128, MI = 2, NJ = 2, PWAVES = TPIX / 64, NT = TPIX * 2;
  __shared__ __attribute__((aligned(16))) unsigned char sW[LIMBS][TCO * ROWB];
  __shared__ __attribute__((aligned(16))) unsigned char sX[LIMBS][TPIX * ROWB];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wco = wid / PWAVES, wpix = wid % PWAVES;
  const unsigned ntiles = (unsigned)a.tiles_co * a.tiles_pix;
  const unsigned logical = gg::xcd_remap(blockIdx.x, ntiles);
  const int tile_co = logical % a.tiles_co, tile_pix = logical / a.tiles_co;
  const int split = blockIdx.y, g = blockIdx.z;
  const int co0 = tile_co * TCO;
  const long long m0 = (long long)tile_pix * TPIX;
  const long long mtot = (long long)a.batch * a.mh * a.mw;
  const int hw = a.h * a.w;

  // ---- gather column: pixel (tid % TPIX), channel half (tid / TPIX) -> 16 consecutive ci of the slab
  const int pcol = tid % TPIX, khalf = tid / TPIX;
  const long long m = m0 + pcol;
  const bool m_ok = m < mtot;
  int pn = 0, base_y = 0, base_x = 0;
  if (m_ok) {
    const int per = a.mh * a.mw;
    pn = (int)(m / per);
    const int rem = (int)(m - (long long)pn * per);
    const int qy = rem / a.mw, qx = rem - qy * a.mw;
    base_y = qy * a.bs + a.byo;
    base_x = qx * a.bs + a.bxo;
  }
  const int chan0 = (pn * a.groups + g) * a.cin_g;
  const float* xg = a.x + (size_t)chan0 * hw;
  const float* sg = IN_SCALE ? a.in_scale + chan0 : nullptr;
  // The gather through ONE buffer resource over the whole input (when it is smaller than 2 GB): lane offset = the
  // pixel's byte offset (its image included: the lanes of a wave may straddle images), scalar offset = the channel.
  // The pointer form spends two 64-bit VALU adds per loaded element on addresses - a fifth of the VALU work of a
  // kernel whose gather is re-done for every tap.
  const long long x_bytes = (long long)a.batch * a.groups * a.cin_g * hw * 4;
  const bool xbuf = x_bytes < (1LL << 31);
  const __amdgpu_buffer_rsrc_t xrs = uniform_rsrc(a.x, xbuf ? (int)x_bytes : 0);
  const int khalf_u = __builtin_amdgcn_readfirstlane(khalf);          // tid / TPIX is wave-uniform (TPIX % 64 == 0)
  // ---- weight rows: row (tid >> 1), 16-k part (tid & 1)
  const int wrow = (tid >> 1) & (TCO - 1), wpart = tid & 1;
  const bool w_thr = tid < 2 * TCO;                 // the first 256 threads move the weight slab
  const bool w_ok = (co0 + wrow) < a.cout_g;
  const int kfull = KS * KS * a.cin_g;
  const unsigned short* wrow_ptr = a.wsplit + ((size_t)g * a.cout_g + (w_ok ? co0 + wrow : 0)) * kfull + wpart * EPT;

  const int cslabs = a.cin_g / BKS;                 // slabs per tap
  const int slab0 = split * a.slabs_per_split;
  int slab1 = slab0 + a.slabs_per_split;
  if (slab1 > a.nslabs) slab1 = a.nslabs;

  float xa[EPT];
  float4 rs4[EPT / 4];
  U4 wv[LIMBS][EPT / 8];
  bool x_ok = false;

  auto load_slab = [&](int slab) {
    const int t = slab / cslabs;
    const int ci0 = (slab - t * cslabs) * BKS;
    int ky, kx, dy, dx;
    if (MODE == 0) {
      ky = t / KS; kx = t - ky * KS; dy = ky; dx = kx;
    } else {
      const int jy = (a.ntx == 2) ? (t >> 1) : t;
      const int jx = (a.ntx == 2) ? (t & 1) : 0;
      ky = a.py + 2 * jy; kx = a.px + 2 * jx; dy = -jy; dx = -jx;
    }
    const int iy = base_y + dy, ix = base_x + dx;
    x_ok = m_ok & ((unsigned)iy < (unsigned)a.h) & ((unsigned)ix < (unsigned)a.w);
    const int cbase = ci0 + khalf * EPT;
    if (xbuf) {
      const unsigned vo = x_ok ? (unsigned)(chan0 * hw + iy * a.w + ix) * 4u : kOobOffset;
      const int so = __builtin_amdgcn_readfirstlane((ci0 + khalf_u * EPT) * hw * 4);
#pragma unroll
      for (int j = 0; j < EPT; ++j) xa[j] = buffer_load_f32(xrs, vo, so + j * hw * 4);
    } else {
      const float* src = xg + (x_ok ? (size_t)cbase * hw + iy * a.w + ix : 0);
      const int step = x_ok ? hw : 0;
#pragma unroll
      for (int j = 0; j < EPT; ++j) xa[j] = src[(size_t)j * step];
    }
    if (IN_SCALE) {
      const float4* s4 = reinterpret_cast<const float4*>(sg + cbase);
#pragma unroll
      for (int j = 0; j < EPT / 4; ++j) rs4[j] = s4[j];
    }
    const unsigned short* wsrc = wrow_ptr + (size_t)(ky * KS + kx) * a.cin_g + ci0;
#pragma unroll
    for (int l = 0; l < LIMBS && w_thr; ++l) {
      const U4* w4 = reinterpret_cast<const U4*>(wsrc + (size_t)l * a.wsplit_stride);
#pragma unroll
      for (int q = 0; q < EPT / 8; ++q) wv[l][q] = w4[q];
    }
  };
  // the gathered slab in its final fp32 form (zero padding, style); binary16 limbs: + this wave's largest magnitude
  // for the block exponent (see BlockExp)
  __shared__ float sAmax[16];
  BlockExp bexp;
  auto prep_slab = [&]() {
    if (!xbuf) {          // (the buffer path's out-of-range offset already returned 0 for masked lanes)
#pragma unroll
      for (int j = 0; j < EPT; ++j) xa[j] = x_ok ? xa[j] : 0.f;
    }
    if (IN_SCALE) {
      const float* rs = reinterpret_cast<const float*>(rs4);
#pragma unroll
      for (int j = 0; j < EPT; ++j) xa[j] *= rs[j];
    }
    if (F16) {
      float m = 0.f;
#pragma unroll
      for (int j = 0; j < EPT; j += 2) m = fmaxf(fmaxf(m, fabsf(xa[j])), fabsf(xa[j + 1]));      // v_max3_f32
      publish_wave_amax(m, sAmax, wid, lane);
    }
  };
  auto store_slab = [&]() {
    float v[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) v[j] = xa[j];
    if (F16 && bexp.e != 0) {                      // uniform branch: most tiles never leave E = 0
      const float ps = exp2i(-bexp.e);
#pragma unroll
      for (int j = 0; j < EPT; ++j) v[j] *= ps;
    }
#pragma unroll
    for (int l = 0; l < LIMBS; ++l) {
      unsigned pk[EPT / 2];
#pragma unroll
      for (int j = 0; j < EPT / 2; ++j) {
        pk[j] = L::pack2(v[2 * j], v[2 * j + 1], l == 0);
        if (l + 1 < LIMBS) {                       // residual for the next limb
          v[2 * j] -= L::lo(pk[j]);
          v[2 * j + 1] -= L::hi(pk[j]);
        }
      }
      U4* dst = reinterpret_cast<U4*>(&sX[l][pcol * ROWB + khalf * EPT * 2]);
      U4* wd = reinterpret_cast<U4*>(&sW[l][wrow * ROWB + wpart * EPT * 2]);
      const U4 z{0u, 0u, 0u, 0u};
#pragma unroll
      for (int q = 0; q < EPT / 8; ++q) {
        dst[q] = U4{pk[4 * q], pk[4 * q + 1], pk[4 * q + 2], pk[4 * q + 3]};
        if (w_thr) wd[q] = w_ok ? wv[l][q] : z;
      }
    }
  };

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (slab0 < slab1) {
    const int kh = lane >> 5, l31 = lane & 31;
    load_slab(slab0);
    if (F16) {
      prep_slab();
      __syncthreads();
    }
    for (int slab = slab0; slab < slab1; ++slab) {
      if (F16) {            // the slab's block exponent (published behind the previous barrier)
        const float f = block_exp_update(bexp, read_block_amax<NT / 64>(sAmax), a.exp_lo);
        if (f != 1.f) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[i][j][r] *= f;
        }
      } else {
        prep_slab();
      }
      store_slab();
      __syncthreads();
      if (slab + 1 < slab1) load_slab(slab + 1);
#pragma unroll
      for (int ks = 0; ks < BKS / 16; ++ks) {
        bf16x8 fa[LIMBS][MI], fb[LIMBS][NJ];
#pragma unroll
        for (int l = 0; l < LIMBS; ++l) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
            fa[l][i] = *reinterpret_cast<const bf16x8*>(&sW[l][((wco * MI + i) * 32 + l31) * ROWB + ks * 32 + kh * 16]);
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            fb[l][j] = *reinterpret_cast<const bf16x8*>(&sX[l][((wpix * NJ + j) * 32 + l31) * ROWB + ks * 32 + kh * 16]);
        }
        // smallest terms first
#pragma unroll
        for (int sum = LIMBS - 1; sum >= 0; --sum)
#pragma unroll
          for (int la = 0; la <= sum; ++la) {
            const int lb = sum - la;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
              for (int j = 0; j < NJ; ++j)
                acc[i][j] = L::mfma(fa[la][i], fb[lb][j], acc[i][j]);
          }
      }
      if (F16 && slab + 1 < slab1) {       // the next slab's loads had this slab's MFMAs to land
        __builtin_amdgcn_sched_barrier(0);
        prep_slab();
      }
      __syncthreads();
    }
  }

  const int ohw = a.oh * a.ow;
  const bool partial = a.part != nullptr;
  const float esc = F16 ? exp2i(bexp.e) : 1.f;          // undo the block exponent (exact)
  float* ybase = partial ? a.part + (size_t)split * a.part_stride : a.y;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const long long mm = m0 + (wpix * NJ + j) * 32 + (lane & 31);
    if (mm >= mtot) continue;
    const int per = a.mh * a.mw;
    const int on = (int)(mm / per);
    const int rem = (int)(mm - (long long)on * per);
    const int qy = rem / a.mw, qx = rem - qy * a.mw;
    const int oy = qy * a.ys + a.yo, ox = qx * a.xs + a.xo;
    const int ochan0 = (on * a.groups + g) * a.cout_g;
    float* yp = ybase + (size_t)ochan0 * ohw + (size_t)oy * a.ow + ox;
    const float* osc = (a.out_scale && !partial) ? a.out_scale + ochan0 : nullptr;
    const float* bia = (a.bias && !partial) ? a.bias + g * a.cout_g : nullptr;
    const float* res = (a.residual && !partial) ? a.residual + (size_t)ochan0 * ohw + (size_t)oy * a.ow + ox : nullptr;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + (wco * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co >= a.cout_g) continue;
        float v = acc[i][j][r] * esc;
        if (!partial) v *= a.acc_scale;
        if (osc) v *= osc[co];
        if (bia) v += bia[co];
        if (res) v += res[(size_t)co * ohw];
        yp[(size_t)co * ohw] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 specialisation of the split-precision kernel with input-patch reuse.
// The generic kernel above re-gathers (and re-splits) every input element once per tap: 9x the L2->CU
// traffic and 9x the conversion work, and at bf16 MFMA rates that traffic is what bounds it
// (measured: ~3.9 TB/s of L2 reads on the 128->128 @256^2 layer).  Here K is ordered (ci-chunk, tap):
// for each chunk of 32 input channels the block stages the halo'd input patch of its 128-pixel tile
// ((TH+2) x (TW+2) pixels, TH*TW = 128) in LDS ONCE, already split into bf16 limbs and scaled by the
// style, and the 9 taps read their B fragments from that patch at shifted pixel offsets.  The next
// chunk's patch is prefetched into registers while the 9 tap slabs of the current chunk run.
// ------------------------------------------------------------------------------------------------
// pixels per tile: 128 (4 waves) or 256 (8 waves: half the weight traffic per output, the dominant L2 stream)
constexpr int patch_pixels(int tpix) { return (tpix / 64 + 2) * 66; }    // (TH+2)*(TW+2) for TW = 64

// MI = 32-row co sub-tiles per wave: 2 -> 128-channel tiles; 1 -> 64-channel tiles (layers with cout <= 64 would
// otherwise spend half of their MFMAs on zero rows)
// TPI = tap slabs staged per barrier interval.  With two or three limbs a tap already carries 24 / 48 MFMAs per wave
// between its two barriers; with ONE limb (plain bf16) it carries 8, and the barrier pair costs about as much as the
// MFMAs - so the single-limb instantiations stage a whole row of taps (ky fixed, kx = 0..2) per interval.
// MASK: 0 = plain; 1 = leaky-ReLU gradient mask on the gathered input from the layer's saved fp32 output (a second
// patch: 32 more loads and prefetch registers per lane and chunk); 2 = the same mask from the 1-bit sign plane the
// forward epilogue wrote (ConvArgs::mask_bits: one word per lane and chunk)
template <int LIMBS, bool IN_SCALE, int TPIX, int MI = 2, int MASK = 0, int TPI = (LIMBS == 1 ? 3 : 1),
          bool F16 = false>
__global__ __launch_bounds__(TPIX * 2, 2) void conv3x3_patch_kernel(const ConvArgs a, int tw_log2) {
  using L = Limb<F16>;
  constexpr int TCO = MI * 64, NJ = 2, NT = TPIX * 2, PWAVES = TPIX / 64;
  static_assert(9 % TPI == 0, "TPI must divide the 9 taps");
  // PIPE: the 8-wave tile (alone on its CU) double-buffers the weight slab and software-pipelines the tap loop - see
  // the main loop.  The 4-wave tiles keep one slab: two of their blocks share a CU and fill each other's stalls.
  constexpr bool PIPE = (TPIX == 256 && TPI == 1);
  constexpr int WBUF = PIPE ? 2 : 1;
  // 3-limb rows are 240 bytes per pixel: 32-wide tiles ((4+2) x (32+2) patch pixels) keep the block under 80 KB of
  // LDS, i.e. two blocks per CU instead of one
  constexpr int PATCH_MAX = LIMBS == 3 ? 6 * 34 : patch_pixels(TPIX);
  // one LDS arena: [limb][patch rows] then [limb][weight rows]; reused as the epilogue staging buffer (32 x 64
  // floats per wave - with a single limb that is the larger of the two uses)
  // (the epilogue also keeps the tile's per-channel scale / bias / activation bias and per-pixel noise in LDS: EPI_BYTES)
  constexpr int MAIN_BYTES = LIMBS * (PATCH_MAX + WBUF * TPI * TCO) * ROWB, STAGE_BYTES = (NT / 64) * 32 * 64 * 4;
  constexpr int EPI_BYTES = (3 * TCO + TPIX) * 4;
  constexpr int SMEM_BYTES = MAIN_BYTES > STAGE_BYTES + EPI_BYTES ? MAIN_BYTES : STAGE_BYTES + EPI_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES + 64];
  float* sAmax = reinterpret_cast<float*>(smem + SMEM_BYTES);      // per-wave operand maxima (binary16 limbs: BlockExp)
  unsigned char (*sP)[PATCH_MAX * ROWB] = reinterpret_cast<unsigned char (*)[PATCH_MAX * ROWB]>(smem);
  unsigned char (*sW)[TCO * ROWB] = reinterpret_cast<unsigned char (*)[TCO * ROWB]>(smem + LIMBS * PATCH_MAX * ROWB);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wco = wid / PWAVES, wpix = wid % PWAVES;
  const int TW = 1 << tw_log2, TH = TPIX >> tw_log2, PW = TW + 2, PP = (TH + 2) * PW;
  const unsigned ntiles = (unsigned)a.tiles_co * a.tiles_pix;
  const unsigned logical = gg::xcd_remap(blockIdx.x, ntiles);
  const int tile_co = logical % a.tiles_co, tile_pix = logical / a.tiles_co;
  const int split = blockIdx.y, g = blockIdx.z;
  const int co0 = tile_co * TCO;
  const int hw = a.h * a.w;
  // tile -> (image, tile row, tile col)
  const int tiles_x = a.w >> tw_log2, tiles_y = a.h / TH;
  const int pn = tile_pix / (tiles_x * tiles_y);
  const int trem = tile_pix - pn * tiles_x * tiles_y;
  const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;

  const int chan0 = (pn * a.groups + g) * a.cin_g;
  const float* sg = IN_SCALE ? a.in_scale + chan0 : nullptr;
  // the image-group's input slab as a buffer resource: the gather is `scalar base + 32-bit lane offset + scalar
  // channel offset` (one address VGPR instead of 32 64-bit pointers), and lanes outside the image use an offset
  // beyond num_records, which the hardware range check turns into 0.0 (the zero padding) without a select
  const __amdgpu_buffer_rsrc_t xr = uniform_rsrc(a.x + (size_t)chan0 * hw, a.cin_g * hw * 4);
  const __amdgpu_buffer_rsrc_t mr = MASK == 1 ? uniform_rsrc(a.mask_ref + (size_t)chan0 * hw, a.cin_g * hw * 4) : xr;
  // sign plane of image pn (groups = 1): word (pixel, chunk) at (pixel * bit_words + chunk) * 4
  const __amdgpu_buffer_rsrc_t br =
      MASK == 2 ? uniform_rsrc(reinterpret_cast<const float*>(a.mask_bits) + (size_t)pn * hw * a.bit_words,
                               hw * a.bit_words * 4)
                : xr;
  const float mpos = a.mask_gain, mneg = a.mask_gain * a.mask_alpha;

  // ---- patch gather: thread -> patch pixel `tid` (all 32 channels of the chunk); for TW = 64 the patch has
  //      264 pixels: the 8 left-over pixels x 32 channels are exactly one extra element per thread
  const bool pin = tid < PP;
  unsigned pvoff;
  {
    const int pr = tid / PW, pc = tid - pr * PW;
    const int iy = y0 + pr - 1, ix = x0 + pc - 1;
    const bool pok = pin & ((unsigned)iy < (unsigned)a.h) & ((unsigned)ix < (unsigned)a.w);
    pvoff = pok ? (unsigned)(iy * a.w + ix) * 4u : kOobOffset;
  }
  const unsigned bvoff = (MASK == 2 && pvoff != kOobOffset) ? pvoff * (unsigned)a.bit_words : kOobOffset;
  const int lpp = NT + (tid & 7), lci = (tid >> 3) & (BKS - 1);     // left-over element (128-pixel tile, TW = 64 only)
  const bool lin = (NT == 256) && lpp < PP;
  unsigned lvoff, lbvoff = kOobOffset;
  {
    const int pr = lpp / PW, pc = lpp - pr * PW;
    const int iy = y0 + pr - 1, ix = x0 + pc - 1;
    const bool lok = lin & ((unsigned)iy < (unsigned)a.h) & ((unsigned)ix < (unsigned)a.w);
    lvoff = lok ? (unsigned)(lci * hw + iy * a.w + ix) * 4u : kOobOffset;
    if (MASK == 2 && lok) lbvoff = (unsigned)(iy * a.w + ix) * 4u * (unsigned)a.bit_words;
  }
  // ---- weight rows
  const int wrow = (tid >> 1) & (TCO - 1), wpart = tid & 1;
  const bool w_thr = tid < 2 * TCO;                       // the first 256 threads move the weight slab
  const bool w_ok = (co0 + wrow) < a.cout_g;
  const int kfull = 9 * a.cin_g;
  // weights through a buffer resource too: loop-invariant lane offset + scalar (tap, chunk, limb) offset, so the
  // loads need no address VGPRs (no WAR wait on the previous slab's registers) and rows beyond cout read as zero
  const __amdgpu_buffer_rsrc_t wr = uniform_rsrc(reinterpret_cast<const float*>(a.wsplit),
                                                 (int)(a.wsplit_stride * 2 * LIMBS));
  const unsigned wvoff = w_ok ? (unsigned)((((size_t)g * a.cout_g + co0 + wrow) * kfull + wpart * EPT) * 2) : kOobOffset;
  const int wlimb = __builtin_amdgcn_readfirstlane((int)(a.wsplit_stride * 2));     // bytes between limb planes

  const int chunk0 = split * a.slabs_per_split;
  int chunk1 = chunk0 + a.slabs_per_split;
  if (chunk1 > a.nslabs) chunk1 = a.nslabs;

  float xa[BKS], xl = 0.f;
  float xm[MASK == 1 ? BKS : 1], xml = 0.f;      // saved activation output at the same positions (MASK 1)
  unsigned mb = 0, mlb = 0;                      // sign words of the lane's pixel / of its left-over element (MASK 2)
  U4 wv[TPI][LIMBS][EPT / 8];

  auto load_bits = [&](int chunk) {
    mb = __builtin_bit_cast(unsigned, buffer_load_f32(br, bvoff, chunk * 4));
    if (NT == 256) mlb = __builtin_bit_cast(unsigned, buffer_load_f32(br, lbvoff, chunk * 4));
  };
  auto load_patch = [&](int chunk) {
    const int cbase = __builtin_amdgcn_readfirstlane(chunk * BKS * hw * 4);
#pragma unroll
    for (int j = 0; j < BKS; ++j) xa[j] = buffer_load_f32(xr, pvoff, cbase + j * hw * 4);
    xl = buffer_load_f32(xr, lvoff, cbase);
    if (MASK == 1) {
#pragma unroll
      for (int j = 0; j < BKS; ++j) xm[MASK == 1 ? j : 0] = buffer_load_f32(mr, pvoff, cbase + j * hw * 4);
      xml = buffer_load_f32(mr, lvoff, cbase);
    }
    if (MASK == 2) load_bits(chunk);
  };
  // The same loads in slices [j0, j1) (+ the left-over element with `tail`): the pipelined tile issues the next chunk's
  // patch a few channels per tap instead of all 33 loads per lane at the top of the chunk.  A CU keeps only a limited
  // number of requests in flight, so the burst held every wave in its VMEM issue phase - in front of the chunk's first
  // MFMA - for 1.4 - 4 thousand cycles per chunk (r03 session O: 7 - 16 % of the launch).
  auto load_patch_slice = [&](int chunk, int j0, int j1, bool tail) {
    const int cbase = __builtin_amdgcn_readfirstlane(chunk * BKS * hw * 4);
#pragma unroll
    for (int j = 0; j < BKS; ++j)
      if (j >= j0 && j < j1) {
        xa[j] = buffer_load_f32(xr, pvoff, cbase + j * hw * 4);
        if (MASK == 1) xm[MASK == 1 ? j : 0] = buffer_load_f32(mr, pvoff, cbase + j * hw * 4);
      }
    if (tail) {
      xl = buffer_load_f32(xr, lvoff, cbase);
      if (MASK == 1) xml = buffer_load_f32(mr, lvoff, cbase);
      if (MASK == 2) load_bits(chunk);
    }
  };
  // prep_patch: the chunk's registers in their final fp32 form (activation mask, style); with binary16 limbs also
  // this wave's largest magnitude, published for the block exponent.  store_patch: scale by 2^-E, split, write.
  BlockExp bexp;
  // (in parts: the pipelined tile finishes channels [j0, j1) per tap while the matrix pipe works, `last` adds the
  // left-over element and publishes)
  float pmax = 0.f;
  auto prep_patch_part = [&](int chunk, int j0, int j1, bool last) {
    if (j0 == 0) pmax = 0.f;
#pragma unroll
    for (int j = 0; j < BKS; ++j)
      if (j >= j0 && j < j1) {
        if (MASK == 1) xa[j] *= xm[MASK == 1 ? j : 0] > 0.f ? mpos : mneg;
        if (MASK == 2) xa[j] *= ((mb >> j) & 1u) ? mpos : mneg;
      }
    if (IN_SCALE && pin) {
#pragma unroll
      for (int j = 0; j < BKS; ++j)
        if (j >= j0 && j < j1) xa[j] *= sg[chunk * BKS + j];
    }
    if (F16) {
#pragma unroll
      for (int j = 0; j < BKS; ++j)
        if (j >= j0 && j < j1) pmax = fmaxf(pmax, fabsf(xa[j]));
    }
    if (last) {
      if (MASK == 1) xl *= xml > 0.f ? mpos : mneg;
      if (MASK == 2) xl *= ((mlb >> lci) & 1u) ? mpos : mneg;
      if (IN_SCALE && lin) xl *= sg[chunk * BKS + lci];
      if (F16) publish_wave_amax(fmaxf(pmax, fabsf(xl)), sAmax, wid, lane);
    }
  };
  auto prep_patch = [&](int chunk) { prep_patch_part(chunk, 0, BKS, true); };
  auto store_patch = [&]() {
    if (F16 && bexp.e != 0) {                      // uniform branch: most tiles never leave E = 0
      const float ps = exp2i(-bexp.e);
#pragma unroll
      for (int j = 0; j < BKS; ++j) xa[j] *= ps;
      xl *= ps;
    }
    if (pin) {
#pragma unroll
      for (int l = 0; l < LIMBS; ++l) {
        U4* dst = reinterpret_cast<U4*>(&sP[l][tid * ROWB]);
#pragma unroll
        for (int q = 0; q < BKS / 8; ++q) {
          unsigned pk[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int j = 8 * q + 2 * e;
            pk[e] = L::pack2(xa[j], xa[j + 1], l == 0);
            if (l + 1 < LIMBS) {
              xa[j] -= L::lo(pk[e]);
              xa[j + 1] -= L::hi(pk[e]);
            }
          }
          dst[q] = U4{pk[0], pk[1], pk[2], pk[3]};
        }
      }
    }
    if (lin) {
      float v = xl;
#pragma unroll
      for (int l = 0; l < LIMBS; ++l) {
        const unsigned short hb = L::one(v, l == 0);
        *reinterpret_cast<unsigned short*>(&sP[l][lpp * ROWB + lci * 2]) = hb;
        v -= L::back(hb);
      }
    }
  };
  // top of a chunk: binary16 limbs take the block exponent published behind the last barrier (and rescale the
  // accumulators if it grew); the other formats finish the registers here, as before
  // interval i of a chunk covers taps [i * TPI, i * TPI + TPI)
  auto load_w = [&](int chunk, int interval) {
    if (!w_thr) return;
#pragma unroll
    for (int u = 0; u < TPI; ++u) {
      const int soff = __builtin_amdgcn_readfirstlane(((interval * TPI + u) * a.cin_g + chunk * BKS) * 2);
#pragma unroll
      for (int l = 0; l < LIMBS; ++l) {
#pragma unroll
        for (int q = 0; q < EPT / 8; ++q)
          wv[u][l][q] = buffer_load_u4(wr, wvoff, soff + l * wlimb + q * 16);
      }
    }
  };
  auto store_w = [&](int buf) {
    if (!w_thr) return;
#pragma unroll
    for (int u = 0; u < TPI; ++u)
#pragma unroll
      for (int l = 0; l < LIMBS; ++l) {
        U4* wd = reinterpret_cast<U4*>(&sW[(buf * TPI + u) * LIMBS + l][wrow * ROWB + wpart * EPT * 2]);
#pragma unroll
        for (int q = 0; q < EPT / 8; ++q) wd[q] = wv[u][l][q];
      }
  };

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int kh = lane >> 5, l31 = lane & 31;
  // lane's pixels inside the tile -> byte offset of the (ky = 0, kx = 0) patch pixel
  int pbase[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int p = (wpix * NJ + j) * 32 + l31;
    const int r = p >> tw_log2, c = p & (TW - 1);
    pbase[j] = (r * PW + c) * ROWB;
  }

  float amax_next = 0.f;                // pipelined tile: the next chunk's magnitude, fetched one barrier early
  auto begin_chunk = [&](int chunk) {
    if (F16) {
      const float f = block_exp_update(bexp, PIPE ? amax_next : read_block_amax<NT / 64>(sAmax), a.exp_lo);
      if (f != 1.f) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] *= f;
      }
    } else {
      prep_patch(chunk);
    }
    store_patch();
  };

  if (chunk0 < chunk1) {
    load_patch(chunk0);
    load_w(chunk0, 0);
    if (F16) prep_patch(chunk0);       // published by the barrier below (PIPE) / the chunk loop's first barrier
    if constexpr (PIPE) {
      if (F16) {
        __syncthreads();
        amax_next = read_block_amax<NT / 64>(sAmax);
      }
      // Software-pipelined tap loop (BKS = 32: two k-steps per tap).  Tap t reads weight buffer t & 1 while slab t + 1
      // is written to the other one, so ONE barrier per tap both publishes slab t + 1 and retires buffer t & 1; the
      // fragments of a tap's first k-step are fetched right after the previous tap's barrier and those of its second
      // k-step before its own, so every LDS fetch has 12 MFMAs (per wave; 24 per SIMD) in front of the instruction
      // that needs it.  With one slab and two barriers per tap the first MFMAs after each barrier waited for the
      // LDS round trip with nothing queued on the matrix pipe: SQ_WAIT_ANY was 32 % of the wave cycles and the pipe
      // 61 % busy inside the main loop.
      typedef bf16x8 FragA[LIMBS][MI];
      typedef bf16x8 FragB[LIMBS][NJ];
      FragA fa0, fa1;
      FragB fb0, fb1;
      auto read_frag = [&](FragA& fa, FragB& fb, int buf, int tapoff, int ks) {
#pragma unroll
        for (int l = 0; l < LIMBS; ++l) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
            fa[l][i] = *reinterpret_cast<const bf16x8*>(
                &sW[buf * LIMBS + l][((wco * MI + i) * 32 + l31) * ROWB + ks * 32 + kh * 16]);
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            fb[l][j] = *reinterpret_cast<const bf16x8*>(&sP[l][pbase[j] + tapoff + ks * 32 + kh * 16]);
        }
      };
      auto mma = [&](const FragA& fa, const FragB& fb) {
#pragma unroll
        for (int sum = LIMBS - 1; sum >= 0; --sum)
#pragma unroll
          for (int la = 0; la <= sum; ++la) {
            const int lb = sum - la;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
              for (int j = 0; j < NJ; ++j)
                acc[i][j] = L::mfma(fa[la][i], fb[lb][j], acc[i][j]);
          }
      };
      for (int chunk = chunk0; chunk < chunk1; ++chunk) {
        // here: every wave is past the barrier that followed its last LDS fetch of the previous chunk (or at kernel
        // start); registers hold this chunk's patch and its tap-0 slab
        if (F16 && MASK == 1 && chunk > chunk0) {
          prep_patch(chunk);
          __syncthreads();
          amax_next = read_block_amax<NT / 64>(sAmax);
        }
        begin_chunk(chunk);
        store_w(0);
        load_w(chunk, 1);
        const bool more = chunk + 1 < chunk1;
        __syncthreads();
        read_frag(fa0, fb0, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int buf = t & 1;
          const int ky = t / 3, kx = t - ky * 3;
          const int tapoff = (ky * PW + kx) * ROWB;
          if (t < 8) {
            store_w(buf ^ 1);                                    // slab t + 1
            if (t + 2 < 9) load_w(chunk, t + 2);
            else if (more) load_w(chunk + 1, 0);
          }
          // next chunk's patch: six channels per tap over taps 0 .. 5 (the last ones land three taps before their use)
          if (more && t < 6) load_patch_slice(chunk + 1, 6 * t, 6 * t + 6 < BKS ? 6 * t + 6 : BKS, t == 5);
          read_frag(fa1, fb1, buf, tapoff, 1);
          // (the scheduler otherwise sinks each fetch down to its first use to shorten the live ranges, which puts
          // the LDS round trip back in front of the MFMAs)
          __builtin_amdgcn_sched_barrier(0);
          // binary16 limbs: the next chunk's patch is complete in registers (issued over taps 0 .. 5) - finish it (style,
          // running maximum) next to tap 8's MFMAs and publish the block's magnitude behind this tap's barrier.
          // Measured alternatives (same box, profiles/r04_e_block_exponent_placement_ab.txt): a third of the chunk per
          // tap over taps 6 .. 8: +1 % on this kernel (longer live ranges); at the chunk top behind an extra barrier: +1 %.
          // The MASKED variant (64 prefetch registers: values + mask references) spills with any early form - it does
          // the work at the chunk top, behind one extra barrier per chunk (-0.26 ms per step against the spilling form).
          // (MASK 2 - the mask from the sign plane, one word per lane - has the plain tile's registers and takes its form)
          if (F16 && more && MASK != 1 && t == 8) prep_patch(chunk + 1);

          mma(fa0, fb0);
          __builtin_amdgcn_sched_barrier(0);
          __syncthreads();
          if (F16 && MASK != 1 && more && t == 8) amax_next = read_block_amax<NT / 64>(sAmax);   // (lands under the next 12 MFMAs)
          if (t < 8) {
            const int t1 = t + 1, ky1 = t1 / 3, kx1 = t1 - ky1 * 3;
            read_frag(fa0, fb0, buf ^ 1, (ky1 * PW + kx1) * ROWB, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          mma(fa1, fb1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
    for (int chunk = chunk0; chunk < chunk1; ++chunk) {
      __syncthreads();                         // previous chunk's readers are done with sP
      begin_chunk(chunk);
      if (chunk + 1 < chunk1) load_patch(chunk + 1);
      for (int iv = 0; iv < 9 / TPI; ++iv) {
        store_w(0);
        __syncthreads();
        // prefetch the next weight slabs (next interval, or interval 0 of the next chunk)
        if (iv + 1 < 9 / TPI) load_w(chunk, iv + 1);
        else if (chunk + 1 < chunk1) load_w(chunk + 1, 0);
#pragma unroll
        for (int u = 0; u < TPI; ++u) {
          const int t = iv * TPI + u;
          const int ky = t / 3, kx = t - ky * 3;
          const int tapoff = (ky * PW + kx) * ROWB;
#pragma unroll
          for (int ks = 0; ks < BKS / 16; ++ks) {
            bf16x8 fa[LIMBS][MI], fb[LIMBS][NJ];
#pragma unroll
            for (int l = 0; l < LIMBS; ++l) {
#pragma unroll
              for (int i = 0; i < MI; ++i)
                fa[l][i] = *reinterpret_cast<const bf16x8*>(
                    &sW[u * LIMBS + l][((wco * MI + i) * 32 + l31) * ROWB + ks * 32 + kh * 16]);
#pragma unroll
              for (int j = 0; j < NJ; ++j)
                fb[l][j] = *reinterpret_cast<const bf16x8*>(&sP[l][pbase[j] + tapoff + ks * 32 + kh * 16]);
            }
#pragma unroll
            for (int sum = LIMBS - 1; sum >= 0; --sum)
#pragma unroll
              for (int la = 0; la <= sum; ++la) {
                const int lb = sum - la;
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                  for (int j = 0; j < NJ; ++j)
                    acc[i][j] = L::mfma(fa[la][i], fb[lb][j], acc[i][j]);
              }
          }
        }
        __syncthreads();                       // sW may be overwritten
      }
      // binary16 limbs: the next chunk's registers landed during the nine taps; published by the loop's top barrier
      if (F16 && chunk + 1 < chunk1) prep_patch(chunk + 1);
    }
    }
  }
  const float esc = F16 ? exp2i(bexp.e) : 1.f;          // undo the block exponent (exact)

  const int ochan0 = (pn * a.groups + g) * a.cout_g;
  const float* osc = a.out_scale ? a.out_scale + ochan0 : nullptr;
  const float* bia = a.bias ? a.bias + g * a.cout_g : nullptr;
  if (a.part) {               // split-K: raw partial sums; splitk_reduce_kernel finishes (scale, bias, activation)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int p = (wpix * NJ + j) * 32 + (lane & 31);
      const int oy = y0 + (p >> tw_log2), ox = x0 + (p & (TW - 1));
      float* yp = a.part + (size_t)split * a.part_stride + (size_t)ochan0 * hw + (size_t)oy * a.w + ox;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + (wco * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (co >= a.cout_g) continue;
          yp[(size_t)co * hw] = acc[i][j][r] * esc;
        }
      }
    }
    return;
  }
  // Direct stores: the accumulator layout gives each lane ONE pixel x 16 channels, i.e. 4-byte stores that
  // reach HBM as partial lines (measured WRITE_SIZE = 4x the tensor).  Transpose each wave's 32co x 64pix
  // sub-tile through LDS and store 16 B per lane: every store instruction writes 4 full 256 B pixel runs.
  __syncthreads();                                          // sP / sW are dead from here on
  float* stage = reinterpret_cast<float*>(smem) + wid * (32 * 64);
  // Everything the epilogue reads from global memory goes to LDS FIRST: VMEM loads and stores share one in-order
  // counter (vmcnt), so a load issued after a store can only be waited for together with that store's
  // acknowledgement - with the per-channel / per-pixel loads inside the store loop every pass paid a store round
  // trip (16 of them per tile; r03 session I: the epilogue was 15 - 29 % of the large layers' launches).
  float* ep_scale = reinterpret_cast<float*>(smem + STAGE_BYTES);        // acc_scale * out_scale[co]
  float* ep_bias = ep_scale + TCO;                                       // bias[co]
  float* ep_abias = ep_bias + TCO;                                       // activation bias[co]
  float* ep_noise = ep_abias + TCO;                                      // noise at the tile's pixels
  for (int c = tid; c < TCO; c += NT) {
    const int co = co0 + c;
    const bool ok = co < a.cout_g;
    ep_scale[c] = (osc && ok) ? a.acc_scale * esc * osc[co] : a.acc_scale * esc;
    ep_bias[c] = (bia && ok) ? bia[co] : 0.f;
    ep_abias[c] = (a.act && a.act_bias && ok) ? a.act_bias[g * a.cout_g + co] : 0.f;
  }
  for (int p = tid; p < TPIX; p += NT) {
    const int oy = y0 + (p >> tw_log2), ox = x0 + (p & (TW - 1));
    ep_noise[p] = (a.act && a.act_noise) ? a.act_noise[(size_t)pn * hw + (size_t)oy * a.w + ox] : 0.f;
  }
  const float anw = (a.act && a.act_noise) ? a.act_noise_w[0] : 0.f;
  __syncthreads();
  // LDS reads are grouped in front of the LDS writes / global stores that follow them: the parameter arrays and the
  // staging rows live in the same LDS array, so the compiler orders every read after the preceding write and waits for it
  // - one LDS round trip per accumulator register and per store in the interleaved form (24 per 32-channel sub-tile)
  const int pq = wpix * 64 + (lane & 15) * 4;                 // this lane's four pixels in every store of the loop below
  const float4 nz = *reinterpret_cast<const float4*>(ep_noise + pq);
  float tmax = 0.f;                         // amax_out: largest stored magnitude seen by this lane
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    float4 sc4[4], bi4[4];
    float abv[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      sc4[q] = *reinterpret_cast<const float4*>(ep_scale + (wco * MI + i) * 32 + 8 * q + 4 * (lane >> 5));
      bi4[q] = *reinterpret_cast<const float4*>(ep_bias + (wco * MI + i) * 32 + 8 * q + 4 * (lane >> 5));
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) abv[it] = ep_abias[(wco * MI + i) * 32 + it * 4 + (lane >> 4)];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const float4 s4 = sc4[r >> 2], b4 = bi4[r >> 2];
      const float sc = (r & 3) == 0 ? s4.x : (r & 3) == 1 ? s4.y : (r & 3) == 2 ? s4.z : s4.w;
      const float bi = (r & 3) == 0 ? b4.x : (r & 3) == 1 ? b4.y : (r & 3) == 2 ? b4.z : b4.w;
#pragma unroll
      for (int j = 0; j < NJ; ++j) stage[row * 64 + j * 32 + (lane & 31)] = acc[i][j][r] * sc + bi;
    }
    wave_lds_sync();                // the staging rows are this wave's own
    float4 v4s[8];
#pragma unroll
    for (int it = 0; it < 8; ++it)
      v4s[it] = *reinterpret_cast<const float4*>(stage + (it * 4 + (lane >> 4)) * 64 + (lane & 15) * 4);
    unsigned sgn[4] = {0u, 0u, 0u, 0u};     // sign bits of the lane's 4 pixels x its 8 channels of this 32-channel block
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = it * 64 + lane;
      const int row = idx >> 4, c4 = idx & 15;
      const int co = co0 + (wco * MI + i) * 32 + row;
      const int p = wpix * 64 + c4 * 4;
      const int oy = y0 + (p >> tw_log2), ox = x0 + (p & (TW - 1));
      if (co < a.cout_g) {
        float4 v4 = v4s[it];
        if (a.act) {             // (NoiseInjection +) bias + leaky ReLU (networks.py:291-298, 344-350; fused_act.py:74-97)
          const float ab = abv[it];
          float t;
          t = v4.x + anw * nz.x + ab; v4.x = (t > 0.f ? t : t * a.act_alpha) * a.act_gain;
          t = v4.y + anw * nz.y + ab; v4.y = (t > 0.f ? t : t * a.act_alpha) * a.act_gain;
          t = v4.z + anw * nz.z + ab; v4.z = (t > 0.f ? t : t * a.act_alpha) * a.act_gain;
          t = v4.w + anw * nz.w + ab; v4.w = (t > 0.f ? t : t * a.act_alpha) * a.act_gain;
        }
        if (a.amax_out) tmax = fmaxf(fmaxf(tmax, fmaxf(fabsf(v4.x), fabsf(v4.y))), fmaxf(fabsf(v4.z), fabsf(v4.w)));
        if (a.sign_bits) {       // exactly the test the backward applies to the STORED value (fused_act.py:33-38)
          sgn[0] |= (v4.x > 0.f ? 1u : 0u) << row;
          sgn[1] |= (v4.y > 0.f ? 1u : 0u) << row;
          sgn[2] |= (v4.z > 0.f ? 1u : 0u) << row;
          sgn[3] |= (v4.w > 0.f ? 1u : 0u) << row;
        }
        float* dst = a.y + (size_t)(ochan0 + co) * hw + (size_t)oy * a.w + ox;
        if (a.nt_store) __builtin_nontemporal_store(f32x4{v4.x, v4.y, v4.z, v4.w}, reinterpret_cast<f32x4*>(dst));
        else *reinterpret_cast<float4*>(dst) = v4;
      }
    }
    if (a.sign_bits) {
      // the four lanes c4, c4 + 16, c4 + 32, c4 + 48 hold channels == 0, 1, 2, 3 (mod 4) of the same four pixels: OR
      // them together (two butterfly steps), then lane (row-lane r, c4) stores the finished word of pixel 4 c4 + r
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sgn[e] |= (unsigned)__shfl_xor((int)sgn[e], 16, 64);
        sgn[e] |= (unsigned)__shfl_xor((int)sgn[e], 32, 64);
      }
      const int r = lane >> 4;
      const unsigned word = r == 0 ? sgn[0] : r == 1 ? sgn[1] : r == 2 ? sgn[2] : sgn[3];
      const int p = wpix * 64 + (lane & 15) * 4 + r;
      const int oy = y0 + (p >> tw_log2), ox = x0 + (p & (TW - 1));
      const int cb = (co0 >> 5) + wco * MI + i;
      if (cb < a.bit_words)
        a.sign_bits[((size_t)pn * hw + (size_t)oy * a.w + ox) * a.bit_words + cb] = word;
    }
    wave_lds_sync();                // the staging rows are this wave's own
  }
  if (a.amax_out) {
    // ONE atomic per tile (a tile lies inside one image), and only when it can raise the value: per-wave atomics of all
    // tiles of an image onto one address serialised in the L2 (+6 % on the 256^2 layer)
    float m = tmax;
    m = dpp_max<0x128, 0xf>(m);
    m = dpp_max<0x124, 0xf>(m);
    m = dpp_max<0x122, 0xf>(m);
    m = dpp_max<0x121, 0xf>(m);
    m = dpp_max<0x142, 0xa>(m);
    m = dpp_max<0x143, 0xc>(m);
    __syncthreads();                        // every wave is done with its staging rows: ep_scale may be reused
    if (lane == 63) ep_scale[wid] = m;
    __syncthreads();
    if (tid == 0) {
      float bm = ep_scale[0];
      for (int q = 1; q < NT / 64; ++q) bm = fmaxf(bm, ep_scale[q]);
      unsigned* dst = reinterpret_cast<unsigned*>(a.amax_out) + pn;
      const unsigned bits = __builtin_bit_cast(unsigned, bm);
      if (bm > 0.f && __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < bits) atomicMax(dst, bits);
    }
  }
}


// ------------------------------------------------------------------------------------------------
// 3x3 / stride 2 TRANSPOSED convolution, all four output parity classes in one pass.
// (StyleGAN2's up-sampling ModulatedConv2d, reference models/stylegan2/networks.py:268-281, and the
// data gradient of every stride-2 convolution.)
//   out[co, 2q + p - pad] = sum_{ci, j} x[ci, q - j] * W[co, ci, k = p + 2j],   p in {0,1}, k < 3
// A tile is TQ "q" positions (TH x TW of the (H+1) x (W+1) q-grid); each 3x3 tap belongs to exactly one
// parity class (ky & 1, kx & 1), so per 32-channel chunk the block stages the (TH+1) x (TW+1) input patch in
// LDS once (split into bf16 limbs, style-scaled) and runs the 9 tap slabs, tap t accumulating into the
// accumulator set of ITS class.  The MFMA work equals a 3x3 convolution at the input resolution - no
// zero-stuffed taps - and the four classes leave through an LDS transpose that interleaves them, so the
// (2H+1)-wide rows are written as contiguous runs.  (The four-launch formulation wrote stride-2 dwords: 3x
// write amplification, and re-gathered the input once per tap.)
// The q-grid has one more row / column than the input; the extra row is an ordinary tile row, the extra
// column is covered by "edge" tiles of shape TQ x 1.
// ------------------------------------------------------------------------------------------------
template <int LIMBS, bool IN_SCALE, int TQ, bool F16 = false>
__global__ __launch_bounds__(TQ * 4, 2) void convT3x3s2_patch_kernel(const ConvArgs a, int tw_log2_in, int tiles_y,
                                                                     int edge_tiles, int pad) {
  using L = Limb<F16>;
  constexpr int TCO = 128, NJ = 2, NT = TQ * 4, PWAVES = TQ / 64;
  constexpr int PATCH_MAX = 2 * TQ + 2;
  // weight slabs staged per barrier interval: the 8-wave variant (alone on its CU) takes a whole row of taps
  // (ky fixed, kx = 0..2) so that each interval carries 36 instead of 12 MFMAs per wave
  constexpr int TPI = (TQ == 128) ? 3 : 1;
  constexpr int MAIN_BYTES = LIMBS * (PATCH_MAX + TPI * TCO) * ROWB, STAGE_BYTES = (NT / 64) * 8 * 128 * 4;
  constexpr int EPI_BYTES = 2 * TCO * 4;          // per-channel scale and bias of the tile (see the epilogue)
  constexpr int SMEM_BYTES = MAIN_BYTES > STAGE_BYTES + EPI_BYTES ? MAIN_BYTES : STAGE_BYTES + EPI_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES + 64];
  float* sAmax = reinterpret_cast<float*>(smem + SMEM_BYTES);      // per-wave operand maxima (binary16 limbs: BlockExp)
  unsigned char (*sP)[PATCH_MAX * ROWB] = reinterpret_cast<unsigned char (*)[PATCH_MAX * ROWB]>(smem);
  unsigned char (*sW)[TCO * ROWB] = reinterpret_cast<unsigned char (*)[TCO * ROWB]>(smem + LIMBS * PATCH_MAX * ROWB);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wco = wid / PWAVES, wpix = wid % PWAVES;
  const unsigned ntiles = (unsigned)a.tiles_co * a.tiles_pix;
  const unsigned logical = gg::xcd_remap(blockIdx.x, ntiles);
  const int tile_co = logical % a.tiles_co, tile_pix = logical / a.tiles_co;
  const int split = blockIdx.y, g = blockIdx.z;
  const int co0 = tile_co * TCO;
  const int hw = a.h * a.w;
  // tile -> (image, q-tile origin, tile shape)
  const int tiles_x = a.w >> tw_log2_in;
  const int interior = tiles_x * tiles_y, per_img = interior + edge_tiles;
  const int pn = tile_pix / per_img;
  const int trem = tile_pix - pn * per_img;
  int tw_log2, y0, x0;
  if (trem < interior) {
    tw_log2 = tw_log2_in;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    y0 = ty * (TQ >> tw_log2);
    x0 = tx << tw_log2;
  } else {
    tw_log2 = 0;
    y0 = (trem - interior) * TQ;
    x0 = a.w;
  }
  const int TW = 1 << tw_log2, TH = TQ >> tw_log2, PW = TW + 1, PP = (TH + 1) * PW;

  const int chan0 = (pn * a.groups + g) * a.cin_g;
  const float* sg = IN_SCALE ? a.in_scale + chan0 : nullptr;
  const __amdgpu_buffer_rsrc_t xr = uniform_rsrc(a.x + (size_t)chan0 * hw, a.cin_g * hw * 4);

  // ---- patch gather: thread -> (patch pixel, 16 of the chunk's 32 channels); the patch origin is (y0-1, x0-1)
  const int pp = tid & (2 * TQ - 1), half = tid / (2 * TQ);
  const bool pin = pp < PP;
  unsigned pvoff;
  {
    const int pr = pp / PW, pc = pp - pr * PW;
    const int iy = y0 + pr - 1, ix = x0 + pc - 1;
    const bool pok = pin & ((unsigned)iy < (unsigned)a.h) & ((unsigned)ix < (unsigned)a.w);
    pvoff = pok ? (unsigned)(half * 16 * hw + iy * a.w + ix) * 4u : kOobOffset;
  }
  const int lpp = 2 * TQ + (tid & 1), lci = (tid >> 1) & (BKS - 1);      // the last two patch pixels
  const bool lin = tid < 2 * BKS && lpp < PP;
  unsigned lvoff;
  {
    const int pr = lpp / PW, pc = lpp - pr * PW;
    const int iy = y0 + pr - 1, ix = x0 + pc - 1;
    const bool lok = lin & ((unsigned)iy < (unsigned)a.h) & ((unsigned)ix < (unsigned)a.w);
    lvoff = lok ? (unsigned)(lci * hw + iy * a.w + ix) * 4u : kOobOffset;
  }
  // ---- weight rows: every thread moves an equal share of each 128-row x 64-byte slab
  constexpr int WPARTS = NT / TCO, WEPT = BKS / WPARTS;
  const int wrow = tid / WPARTS, wpart = tid % WPARTS;
  const bool w_ok = (co0 + wrow) < a.cout_g;
  const int kfull = 9 * a.cin_g;
  // weights through a buffer resource too: loop-invariant lane offset + scalar (tap, chunk, limb) offset, so the
  // loads need no address VGPRs (no WAR wait on the previous slab's registers) and rows beyond cout read as zero
  const __amdgpu_buffer_rsrc_t wr = uniform_rsrc(reinterpret_cast<const float*>(a.wsplit),
                                                 (int)(a.wsplit_stride * 2 * LIMBS));
  const unsigned wvoff = w_ok ? (unsigned)((((size_t)g * a.cout_g + co0 + wrow) * kfull + wpart * WEPT) * 2) : kOobOffset;
  const int wlimb = __builtin_amdgcn_readfirstlane((int)(a.wsplit_stride * 2));     // bytes between limb planes

  const int chunk0 = split * a.slabs_per_split;
  int chunk1 = chunk0 + a.slabs_per_split;
  if (chunk1 > a.nslabs) chunk1 = a.nslabs;

  float xa[16], xl = 0.f;
  U4 wv[TPI][LIMBS][WEPT / 8];

  auto load_patch = [&](int chunk) {
    const int cbase = __builtin_amdgcn_readfirstlane(chunk * BKS * hw * 4);
#pragma unroll
    for (int j = 0; j < 16; ++j) xa[j] = buffer_load_f32(xr, pvoff, cbase + j * hw * 4);
    xl = buffer_load_f32(xr, lvoff, cbase);
  };
  // sliced form of load_patch (see conv3x3_patch_kernel): channels [j0, j1) of the lane's 16 (+ the left-over element)
  auto load_patch_slice = [&](int chunk, int j0, int j1, bool tail) {
    const int cbase = __builtin_amdgcn_readfirstlane(chunk * BKS * hw * 4);
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j >= j0 && j < j1) xa[j] = buffer_load_f32(xr, pvoff, cbase + j * hw * 4);
    if (tail) xl = buffer_load_f32(xr, lvoff, cbase);
  };
  // prep_patch / store_patch: as in conv3x3_patch_kernel (style first; binary16 limbs publish the wave's magnitude for
  // the block exponent, then scale by 2^-E; split; write)
  BlockExp bexp;
  auto prep_patch = [&](int chunk) {
    if (IN_SCALE) {
      if (pin) {
#pragma unroll
        for (int j = 0; j < 16; ++j) xa[j] *= sg[chunk * BKS + half * 16 + j];
      }
      if (lin) xl *= sg[chunk * BKS + lci];
    }
    if (F16) {
      float m = fabsf(xl);
#pragma unroll
      for (int j = 0; j < 16; ++j) m = fmaxf(m, fabsf(xa[j]));
      publish_wave_amax(m, sAmax, wid, lane);
    }
  };
  auto store_patch = [&]() {
    if (F16 && bexp.e != 0) {
      const float ps = exp2i(-bexp.e);
#pragma unroll
      for (int j = 0; j < 16; ++j) xa[j] *= ps;
      xl *= ps;
    }
    if (pin) {
#pragma unroll
      for (int l = 0; l < LIMBS; ++l) {
        U4* dst = reinterpret_cast<U4*>(&sP[l][pp * ROWB + half * 32]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          unsigned pk[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int j = 8 * q + 2 * e;
            pk[e] = L::pack2(xa[j], xa[j + 1], l == 0);
            if (l + 1 < LIMBS) {
              xa[j] -= L::lo(pk[e]);
              xa[j + 1] -= L::hi(pk[e]);
            }
          }
          dst[q] = U4{pk[0], pk[1], pk[2], pk[3]};
        }
      }
    }
    if (lin) {
      float v = xl;
#pragma unroll
      for (int l = 0; l < LIMBS; ++l) {
        const unsigned short hb = L::one(v, l == 0);
        *reinterpret_cast<unsigned short*>(&sP[l][lpp * ROWB + lci * 2]) = hb;
        v -= L::back(hb);
      }
    }
  };
  // interval i of a chunk covers taps [i * TPI, i * TPI + TPI)
  auto load_w = [&](int chunk, int interval) {
#pragma unroll
    for (int u = 0; u < TPI; ++u) {
      const int soff = __builtin_amdgcn_readfirstlane(((interval * TPI + u) * a.cin_g + chunk * BKS) * 2);
#pragma unroll
      for (int l = 0; l < LIMBS; ++l)
#pragma unroll
        for (int q = 0; q < WEPT / 8; ++q) wv[u][l][q] = buffer_load_u4(wr, wvoff, soff + l * wlimb + q * 16);
    }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int u = 0; u < TPI; ++u)
#pragma unroll
      for (int l = 0; l < LIMBS; ++l) {
        U4* wd = reinterpret_cast<U4*>(&sW[u * LIMBS + l][wrow * ROWB + wpart * WEPT * 2]);
#pragma unroll
        for (int q = 0; q < WEPT / 8; ++q) wd[q] = wv[u][l][q];
      }
  };

  f32x16 acc[4][NJ];                        // [parity class py*2+px][pixel sub-tile]
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][j][r] = 0.f;

  const int kh = lane >> 5, l31 = lane & 31;
  int pbase[NJ];                            // byte offset of the lane's pixels relative to the patch origin
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int p = (wpix * NJ + j) * 32 + l31;
    const int r = p >> tw_log2, c = p & (TW - 1);
    pbase[j] = (r * PW + c) * ROWB;
  }

  if (chunk0 < chunk1) {
    load_patch(chunk0);
    load_w(chunk0, 0);
    if (F16) prep_patch(chunk0);             // published by the chunk loop's first barrier
    for (int chunk = chunk0; chunk < chunk1; ++chunk) {
      __syncthreads();
      if (F16) {          // block exponent of this chunk (rescales the accumulators if it grew)
        const float f = block_exp_update(bexp, read_block_amax<NT / 64>(sAmax), a.exp_lo);
        if (f != 1.f) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[c][j][r] *= f;
        }
      } else {
        prep_patch(chunk);
      }
      store_patch();
      const bool more = chunk + 1 < chunk1;
      // 8-wave tile: the next chunk's patch is issued in slices inside the first two intervals (below); the 4-wave
      // tile (two blocks per CU, one tap per interval) keeps the single burst
      if (more && TPI == 1) load_patch(chunk + 1);
#pragma unroll
      for (int iv = 0; iv < 9 / TPI; ++iv) {
        store_w();
        __syncthreads();
        if (iv + 1 < 9 / TPI) load_w(chunk, iv + 1);
        else if (more) load_w(chunk + 1, 0);
        // (tap, k-step) units of the interval, software-pipelined: the fragments of unit q + 1 are fetched before the
        // MFMAs of unit q, so that an LDS round trip is covered by 6 (two limbs) MFMAs per wave instead of being
        // waited for in front of them (with 255 registers in use the scheduler otherwise fetches just in time)
        // (three limbs: 9 fragments per unit - two sets no longer fit the register file, fetch in place)
        constexpr int UNITS = TPI * (BKS / 16);
        constexpr bool PF = LIMBS <= 2;
        bf16x8 fa[PF ? 2 : 1][LIMBS], fb[PF ? 2 : 1][LIMBS][NJ];
        auto fetch = [&](int slot, int q) {
          const int u = q / (BKS / 16), ks = q % (BKS / 16);
          const int t = iv * TPI + u;
          const int ky = t / 3, kx = t - ky * 3;
          // x[q - j]: patch row/col (q - y0) + 1 - j
          const int tapoff = ((1 - (ky >> 1)) * PW + (1 - (kx >> 1))) * ROWB;
#pragma unroll
          for (int l = 0; l < LIMBS; ++l) {
            fa[slot][l] =
                *reinterpret_cast<const bf16x8*>(&sW[u * LIMBS + l][(wco * 32 + l31) * ROWB + ks * 32 + kh * 16]);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              fb[slot][l][j] = *reinterpret_cast<const bf16x8*>(&sP[l][pbase[j] + tapoff + ks * 32 + kh * 16]);
          }
        };
        if (PF) fetch(0, 0);
#pragma unroll
        for (int q = 0; q < UNITS; ++q) {
          const int slot = PF ? (q & 1) : 0;
          if (!PF) fetch(0, q);
          else if (q + 1 < UNITS) fetch(slot ^ 1, q + 1);
          if (TPI == 3 && more) {            // 2 channels per unit in interval 0, 1 per unit in the first 4 units of interval 1
            if (iv == 0) load_patch_slice(chunk + 1, 2 * q, 2 * q + 2, false);
            else if (iv == 1 && q < 4) load_patch_slice(chunk + 1, 12 + q, 13 + q, q == 3);
          }
          if (PF) __builtin_amdgcn_sched_barrier(0);
          const int t = iv * TPI + q / (BKS / 16);
          const int ky = t / 3, kx = t - ky * 3;
          const int cls = (ky & 1) * 2 + (kx & 1);
#pragma unroll
          for (int sum = LIMBS - 1; sum >= 0; --sum)
#pragma unroll
            for (int la = 0; la <= sum; ++la) {
              const int lb = sum - la;
#pragma unroll
              for (int j = 0; j < NJ; ++j)
                acc[cls][j] = L::mfma(fa[slot][la], fb[slot][lb][j], acc[cls][j]);
            }
          if (PF) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
      }
      // binary16 limbs: the next chunk's registers landed during the nine taps; published by the loop's top barrier
      // (inside the last interval's MFMA region the extra live registers spill: 80 bytes per lane)
      if (F16 && more) prep_patch(chunk + 1);
    }
  }
  const float esc = F16 ? exp2i(bexp.e) : 1.f;          // undo the block exponent (exact)

  const int ohw = a.oh * a.ow;
  const int owp = a.ow, ohwp = ohw;
  const int ochan0 = (pn * a.groups + g) * a.cout_g;
  const float* osc = a.out_scale ? a.out_scale + ochan0 : nullptr;
  const float* bia = a.bias ? a.bias + g * a.cout_g : nullptr;
  if (a.part) {               // split-K: raw partial sums; splitk_reduce_kernel finishes (scale, bias)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int p = (wpix * NJ + j) * 32 + l31;
        const int oy = 2 * (y0 + (p >> tw_log2)) + (c >> 1) - pad, ox = 2 * (x0 + (p & (TW - 1))) + (c & 1) - pad;
        if ((unsigned)oy >= (unsigned)a.oh || (unsigned)ox >= (unsigned)a.ow) continue;
        float* yp = a.part + (size_t)split * a.part_stride + (size_t)ochan0 * ohw + (size_t)oy * a.ow + ox;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (co >= a.cout_g) continue;
          yp[(size_t)co * ohw] = acc[c][j][r] * esc;
        }
      }
    }
    return;
  }
  // Interleave the classes through LDS: per pass 8 channels x (2*64) outputs of one output-row parity.  Each wave
  // owns its staging rows, so passes are ordered by wave-level fences only (no block barriers: waves drift apart and
  // one wave's stores overlap another's LDS traffic).  The two column classes of a q position are adjacent in the
  // output row: one 8-byte LDS write per lane, 16-byte reads, and 16-byte buffer stores (dword-aligned: the rows of
  // the (2W+1)-wide output start at odd offsets) - 4x fewer store instructions than the per-dword form.
  __syncthreads();                                          // sP / sW are dead from here on
  float* stage = reinterpret_cast<float*>(smem) + wid * (8 * 128);
  // per-channel scale / bias through LDS, loaded before the first store (a VMEM load issued after a store is waited
  // for together with that store's acknowledgement - see conv3x3_patch_kernel's epilogue)
  float* ep_scale = reinterpret_cast<float*>(smem + STAGE_BYTES);
  float* ep_bias = ep_scale + TCO;
  for (int c = tid; c < TCO; c += NT) {
    const int co = co0 + c;
    const bool ok = co < a.cout_g;
    ep_scale[c] = (osc && ok) ? a.acc_scale * esc * osc[co] : a.acc_scale * esc;
    ep_bias[c] = (bia && ok) ? bia[co] : 0.f;
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t yr = uniform_rsrc(a.y + (size_t)ochan0 * ohw, a.cout_g * ohw * 4);
  const bool vec = tw_log2 > 0;                             // edge tiles (one q column): scalar stores
  // the lane's 16 channel scales / biases in registers before the passes (LDS reads interleaved with the staging writes
  // are ordered after them by the compiler: one round trip each)
  float4 sc4[4], bi4[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    sc4[q] = *reinterpret_cast<const float4*>(ep_scale + wco * 32 + 8 * q + 4 * (lane >> 5));
    bi4[q] = *reinterpret_cast<const float4*>(ep_bias + wco * 32 + 8 * q + 4 * (lane >> 5));
  }
#pragma unroll
  for (int py = 0; py < 2; ++py) {
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int r = q4 * 4 + rr;
        const int lrow = rr + 4 * (lane >> 5);
        const float sc = rr == 0 ? sc4[q4].x : rr == 1 ? sc4[q4].y : rr == 2 ? sc4[q4].z : sc4[q4].w;
        const float bi = rr == 0 ? bi4[q4].x : rr == 1 ? bi4[q4].y : rr == 2 ? bi4[q4].z : bi4[q4].w;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          float2 v2;
          v2.x = acc[py * 2 + 0][j][r] * sc + bi;
          v2.y = acc[py * 2 + 1][j][r] * sc + bi;
          *reinterpret_cast<float2*>(stage + lrow * 128 + (j * 32 + l31) * 2) = v2;
        }
      }
      wave_lds_sync();
      if (vec) {
        // the pass's four LDS reads first, then its four stores (one LDS round trip per pass instead of four)
        f32x4 v4s[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int idx = it * 64 + lane;
          v4s[it] = *reinterpret_cast<const f32x4*>(stage + (idx >> 5) * 128 + (idx & 31) * 4);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int idx = it * 64 + lane;
          const int lrow = idx >> 5, s = (idx & 31) * 4;
          const f32x4 v4 = v4s[it];
          const int p = wpix * 64 + (s >> 1);
          const int oy = 2 * (y0 + (p >> tw_log2)) + py - pad, ox = 2 * (x0 + (p & (TW - 1))) - pad;
          const int co = co0 + wco * 32 + lrow + 8 * q4;
          const bool rowok = co < a.cout_g && (unsigned)oy < (unsigned)a.oh;
          const unsigned off = (unsigned)(co * ohwp + oy * owp + ox) * 4u;
          if (rowok && ox >= 0 && ox + 3 < a.ow) {
            if (a.nt_store) buffer_store_f32x4_nt(v4, yr, off, 0);
            else buffer_store_f32x4(v4, yr, off, 0);
          } else if (rowok) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              buffer_store_f32(v4[e], yr, (unsigned)(ox + e) < (unsigned)a.ow ? off + 4u * e : kOobOffset, 0);
          }
        }
      } else {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
          const int idx = it * 64 + lane;
          const int lrow = idx >> 7, s = idx & 127;
          const int p = wpix * 64 + (s >> 1);
          const int oy = 2 * (y0 + (p >> tw_log2)) + py - pad, ox = 2 * (x0 + (p & (TW - 1))) + (s & 1) - pad;
          const int co = co0 + wco * 32 + lrow + 8 * q4;
          if (co < a.cout_g && (unsigned)oy < (unsigned)a.oh && (unsigned)ox < (unsigned)a.ow)
            a.y[(size_t)(ochan0 + co) * ohw + (size_t)oy * a.ow + ox] = stage[lrow * 128 + s];
        }
      }
      wave_lds_sync();
    }
  }
}


// wl[limb][g][c][(tap, r)], r (reduction channel) fastest.  `first` / `stride`: this thread's grid-stride walk.
__device__ __forceinline__ void pack_weight_split_body(unsigned short* __restrict__ wl, const float* __restrict__ w,
                                                       long long total, long long limb_stride, int cout_g, int cin_g,
                                                       int kh, int kw, int transpose_io, int flip, float scale,
                                                       int limbs, long long first, long long stride) {
  const int kk = kh * kw;
  for (long long o = first; o < total; o += stride) {
    const int r = (int)(o % cin_g);
    long long q = o / cin_g;
    const int tap = (int)(q % kk);
    q /= kk;
    const int c = (int)(q % cout_g);
    const int g = (int)(q / cout_g);
    int ky = tap / kw, kx = tap % kw;
    if (flip) { ky = kh - 1 - ky; kx = kw - 1 - kx; }
    const size_t src = transpose_io ? (((size_t)g * cin_g + r) * cout_g + c) * kk + ky * kw + kx
                                    : (((size_t)g * cout_g + c) * cin_g + r) * kk + ky * kw + kx;
    float v = w[src] * scale;
    if (limbs & 16) {                    // binary16 limbs (forward packs of the fp16x3 mode), see Limb<true>
      v *= kF16WeightScale;
      for (int l = 0; l < (limbs & 15); ++l) {
        const unsigned short h = f16_limb_rn(v);
        wl[(size_t)l * limb_stride + o] = h;
        v -= Limb<true>::back(h);
      }
      continue;
    }
    for (int l = 0; l < limbs; ++l) {
      const __bf16 h = (__bf16)v;
      wl[(size_t)l * limb_stride + o] = __builtin_bit_cast(unsigned short, h);
      v -= (float)h;
    }
  }
}

__global__ __launch_bounds__(256) void pack_weight_split_kernel(unsigned short* __restrict__ wl,
                                                                const float* __restrict__ w, long long total,
                                                                long long limb_stride, int cout_g, int cin_g, int kh,
                                                                int kw, int transpose_io, int flip, float scale,
                                                                int limbs) {
  pack_weight_split_body(wl, w, total, limb_stride, cout_g, cin_g, kh, kw, transpose_io, flip, scale, limbs,
                         (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x);
}

// Many weights in one launch (the trainer re-packs every trainable convolution weight after each optimizer step):
// blockIdx.y = job, blockIdx.x = one of PACK_MANY_BLOCKS grid-stride workers.
struct PackJob {
  void* dst;
  const float* src;
  long long total;
  long long limb_stride;
  int cout_g, cin_g, kh, kw, transpose_io, flip;
  int limbs;            // 0: fp32 GEMM layout (pack_weight_kernel); 1 | 2 | 3: bf16 limb planes; 18: two binary16 limbs
  float scale;
};
constexpr int PACK_MANY_BLOCKS = 512;        // grid-stride workers per job (small jobs: most exit at once)

__device__ __forceinline__ void pack_weight_body(float* __restrict__ wmat, const float* __restrict__ w, long long total,
                                                 int cout_g, int cin_g, int kh, int kw, int transpose_io, int flip,
                                                 float scale, long long first, long long stride);

// LDS-staged form of pack_weight_split_body for the one-launch re-pack: the element-wise kernel reads the source
// with a stride of kk (or cout_g * kk, transposed layout) floats between neighbouring lanes and leans on the caches
// for the rest of each line (1.4 TB/s over the 43 M trainable weights).  Here a block stages a tile with coalesced
// reads and writes whole runs of the destination:
//   plain layout      tile = one output channel c: cin_g * kk contiguous source floats -> kk rows of cin_g bf16
//   transposed layout tile = 64 reduction channels r x 16 output channels c: per r a run of 16 * kk source floats
//                     -> per (c, tap) a run of 64 bf16
// Odd row lengths in LDS (kk = 9 / 1, 16 * kk + 1) keep the transposing reads conflict-free.
constexpr int PACK_LDS_FLOATS = 64 * (16 * 9 + 1);          // 9280 floats = 37 KB
constexpr int PACK_RT = 64, PACK_CT = 16;

__device__ __forceinline__ bool pack_tiled_ok(const PackJob& j) {
  const int kk = j.kh * j.kw;
  if (!j.limbs || kk > 9) return false;
  return j.transpose_io ? true : (long long)j.cin_g * kk <= PACK_LDS_FLOATS;
}

__device__ __forceinline__ void store_limbs(unsigned short* __restrict__ dst, long long limb_stride, int limbs, float v) {
  if (limbs & 16) {                      // binary16 limbs, pre-scaled (see pack_weight_split_body)
    v *= kF16WeightScale;
    for (int l = 0; l < (limbs & 15); ++l) {
      const unsigned short h = f16_limb_rn(v);
      dst[(size_t)l * limb_stride] = h;
      v -= Limb<true>::back(h);
    }
    return;
  }
  for (int l = 0; l < limbs; ++l) {
    const __bf16 h = (__bf16)v;
    dst[(size_t)l * limb_stride] = __builtin_bit_cast(unsigned short, h);
    v -= (float)h;
  }
}
// two neighbouring reduction channels per lane: one 4-byte store per limb (dst 4-byte aligned: even cin_g, even r,
// even limb_stride)
__device__ __forceinline__ void store_limbs2(unsigned short* __restrict__ dst, long long limb_stride, int limbs, float a,
                                             float b) {
  if (limbs & 16) {
    a *= kF16WeightScale;
    b *= kF16WeightScale;
    for (int l = 0; l < (limbs & 15); ++l) {
      const unsigned pk = (unsigned)f16_limb_rn(a) | ((unsigned)f16_limb_rn(b) << 16);
      *reinterpret_cast<unsigned*>(dst + (size_t)l * limb_stride) = pk;
      a -= Limb<true>::lo(pk);
      b -= Limb<true>::hi(pk);
    }
    return;
  }
  for (int l = 0; l < limbs; ++l) {
    const unsigned pk = pack_bf16x2(a, b);
    *reinterpret_cast<unsigned*>(dst + (size_t)l * limb_stride) = pk;
    a -= bf16_lo(pk);
    b -= bf16_hi(pk);
  }
}

__device__ void pack_weight_split_tiled(const PackJob& j, float* __restrict__ lds, int block, int nblocks) {
  const int kk = j.kh * j.kw, tid = threadIdx.x;
  const int groups = (int)(j.total / ((long long)j.cout_g * j.cin_g * kk));
  unsigned short* wl = reinterpret_cast<unsigned short*>(j.dst);
  const bool pairs = (j.cin_g & 1) == 0 && (j.limb_stride & 1) == 0 && (reinterpret_cast<uintptr_t>(wl) & 3) == 0;
  if (!j.transpose_io) {
    const int ntiles = groups * j.cout_g, run = j.cin_g * kk;
    for (int tile = block; tile < ntiles; tile += nblocks) {
      const float* src = j.src + (size_t)tile * run;          // tile = g * cout_g + c
      for (int i = tid; i < run; i += 256) lds[i] = src[i] * j.scale;
      __syncthreads();
      unsigned short* dst = wl + (size_t)tile * run;
      if (pairs) {
        for (int o = 2 * tid; o < run; o += 512) {            // o = tap' * cin_g + r, r even
          const int tap = o / j.cin_g, r = o - tap * j.cin_g;
          int ky = tap / j.kw, kx = tap - ky * j.kw;
          if (j.flip) { ky = j.kh - 1 - ky; kx = j.kw - 1 - kx; }
          const int t = ky * j.kw + kx;
          store_limbs2(dst + o, j.limb_stride, j.limbs, lds[r * kk + t], lds[(r + 1) * kk + t]);
        }
      } else {
        for (int o = tid; o < run; o += 256) {
          const int tap = o / j.cin_g, r = o - tap * j.cin_g;
          int ky = tap / j.kw, kx = tap - ky * j.kw;
          if (j.flip) { ky = j.kh - 1 - ky; kx = j.kw - 1 - kx; }
          store_limbs(dst + o, j.limb_stride, j.limbs, lds[r * kk + ky * j.kw + kx]);
        }
      }
      __syncthreads();
    }
    return;
  }
  const int LD = PACK_CT * kk + 1;
  const int rt = (j.cin_g + PACK_RT - 1) / PACK_RT, ct = (j.cout_g + PACK_CT - 1) / PACK_CT;
  const int ntiles = groups * rt * ct;
  for (int tile = block; tile < ntiles; tile += nblocks) {
    const int g = tile / (rt * ct), rem = tile - g * rt * ct;
    const int r0 = (rem / ct) * PACK_RT, c0 = (rem % ct) * PACK_CT;
    const int nr = min(PACK_RT, j.cin_g - r0), nc = min(PACK_CT, j.cout_g - c0);
    const int run = nc * kk;
    // source: ((g * cin_g + r) * cout_g + c) * kk + tap
    for (int i = tid; i < nr * run; i += 256) {
      const int r = i / run, e = i - r * run;
      lds[r * LD + e] = j.src[((size_t)(g * j.cin_g + r0 + r) * j.cout_g + c0) * kk + e] * j.scale;
    }
    __syncthreads();
    // destination: ((g * cout_g + c) * kk + tap') * cin_g + r
    if (pairs) {
      for (int o = tid; o < nc * kk * (PACK_RT / 2); o += 256) {
        const int r = (o & (PACK_RT / 2 - 1)) * 2, ct_ = o >> 5;   // ct_ = c * kk + tap'
        if (r >= nr) continue;                                  // (nr is even: cin_g is)
        const int c = ct_ / kk, tap = ct_ - c * kk;
        int ky = tap / j.kw, kx = tap - ky * j.kw;
        if (j.flip) { ky = j.kh - 1 - ky; kx = j.kw - 1 - kx; }
        const int e = c * kk + ky * j.kw + kx;
        store_limbs2(wl + ((size_t)(g * j.cout_g + c0 + c) * kk + tap) * j.cin_g + r0 + r, j.limb_stride, j.limbs,
                     lds[r * LD + e], lds[(r + 1) * LD + e]);
      }
    } else {
      for (int o = tid; o < nc * kk * PACK_RT; o += 256) {
        const int r = o & (PACK_RT - 1), ct_ = o >> 6;
        if (r >= nr) continue;
        const int c = ct_ / kk, tap = ct_ - c * kk;
        int ky = tap / j.kw, kx = tap - ky * j.kw;
        if (j.flip) { ky = j.kh - 1 - ky; kx = j.kw - 1 - kx; }
        store_limbs(wl + ((size_t)(g * j.cout_g + c0 + c) * kk + tap) * j.cin_g + r0 + r, j.limb_stride, j.limbs,
                    lds[r * LD + c * kk + ky * j.kw + kx]);
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void pack_weight_many_kernel(const PackJob* __restrict__ jobs) {
  __shared__ float lds[PACK_LDS_FLOATS];
  const PackJob j = jobs[blockIdx.y];
  const long long first = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)gridDim.x * 256;
  if (pack_tiled_ok(j))
    pack_weight_split_tiled(j, lds, (int)blockIdx.x, (int)gridDim.x);
  else if (j.limbs)
    pack_weight_split_body(reinterpret_cast<unsigned short*>(j.dst), j.src, j.total, j.limb_stride, j.cout_g, j.cin_g,
                           j.kh, j.kw, j.transpose_io, j.flip, j.scale, j.limbs, first, stride);
  else
    pack_weight_body(reinterpret_cast<float*>(j.dst), j.src, j.total, j.cout_g, j.cin_g, j.kh, j.kw, j.transpose_io,
                     j.flip, j.scale, first, stride);
}

// ------------------------------------------------------------------------------------------------
// weight gradient:  dW[co, j] = sum_{pix} dy[n, co, pix] * x[n, ci(j), pix*stride + tap(j) - pad]
//   D[i = co][j = (ci,ky,kx)], reduction index k = pixel.  Both global operands are contiguous
//   along the pixel axis, so they are loaded lane-along-k (128 B rows) and stored to LDS as
//   [k][row] with an odd row stride (129): lane-along-k writes and lane-along-row fragment reads
//   are both conflict free.  K (= N*OH*OW, up to 10^6) is split across blockIdx.y; partial tiles go to the
//   stream's scratch and are summed in split order by a reduce pass (no float atomics: reproducible).
// ------------------------------------------------------------------------------------------------
constexpr int WBK = 32;
constexpr int WT = 128;          // tile edge (co and j)
constexpr int WLD = WT + 1;

// (struct WgradArgs: conv_common.h)

template <int KS>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
  constexpr int KK = KS * KS;
  __shared__ float sD[2][WBK][WLD];   // dy  [k][co]
  __shared__ float sXg[2][WBK][WLD];  // x    [k][j]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wco = wid >> 1, wj = wid & 1;
  const unsigned ntiles = (unsigned)a.tiles_co * a.tiles_j;
  const unsigned logical = gg::xcd_remap(blockIdx.x, ntiles);
  const int tile_co = logical % a.tiles_co, tile_j = logical / a.tiles_co;
  const int g = blockIdx.z;
  const int co0 = tile_co * WT, j0 = tile_j * WT;
  const long long kbeg = (long long)blockIdx.y * a.k_per_split;
  long long kend = kbeg + a.k_per_split;
  if (kend > a.ktot) kend = a.ktot;
  const int ohw = a.oh * a.ow, hw = a.h * a.w;

  const int kl = tid & 31;        // this thread's k (pixel) lane inside a slab
  const int r0 = tid >> 5;        // first row (co / j) it loads; rows r0 + 8*p, p < 16

  // decode of this thread's 16 j columns is loop invariant
  int jci[16], jdy[16], jdx[16];
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const int j = j0 + r0 + 8 * p;
    if (j < a.jtot) {
      const int ci = j / KK, tap = j - ci * KK;
      jci[p] = ci;
      jdy[p] = tap / KS - a.pad;
      jdx[p] = tap % KS - a.pad;
    } else {
      jci[p] = -1; jdy[p] = 0; jdx[p] = 0;
    }
  }

  float rd[16], rx[16];
  unsigned mask_d = 0, mask_x = 0;
  // (image, position inside it) of this thread's pixel, advanced by one slab per call (no 64-bit division per slab)
  int kn;
  unsigned krem;
  {
    const long long k = kbeg + kl;
    kn = (int)(k / ohw);
    krem = (unsigned)(k - (long long)kn * ohw);
  }
  auto load_slab = [&](long long k0) {
    const long long k = k0 + kl;
    const bool ok = k < kend;
    int n = 0, oy = 0, ox = 0;
    if (ok) {
      n = kn;
      oy = (int)(krem / (unsigned)a.ow);
      ox = (int)(krem - (unsigned)oy * (unsigned)a.ow);
    }
    krem += WBK;
    while (krem >= (unsigned)ohw) { krem -= (unsigned)ohw; ++kn; }
    const float* dyp = a.dy + ((size_t)(n * a.groups + g) * a.cout_g) * ohw + (size_t)oy * a.ow + ox;
    const float* xp = a.x + ((size_t)(n * a.groups + g) * a.cin_g) * hw;
    const int by = oy * a.stride, bx = ox * a.stride;
    mask_d = 0;
    mask_x = 0;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int co = co0 + r0 + 8 * p;
      const unsigned okd = (unsigned)ok & (unsigned)(co < a.cout_g);
      rd[p] = dyp[(size_t)co * ohw * okd];
      mask_d |= okd << p;
      const int iy = by + jdy[p], ix = bx + jdx[p];
      const unsigned okx = (unsigned)ok & (unsigned)(jci[p] >= 0) & (unsigned)((unsigned)iy < (unsigned)a.h) &
                           (unsigned)((unsigned)ix < (unsigned)a.w);
      rx[p] = xp[(size_t)((jci[p] * hw + iy * a.w + ix) * (int)okx)];
      mask_x |= okx << p;
    }
  };
  auto store_slab = [&](int buf) {
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      sD[buf][kl][r0 + 8 * p] = ((mask_d >> p) & 1u) ? rd[p] : 0.f;
      sXg[buf][kl][r0 + 8 * p] = ((mask_x >> p) & 1u) ? rx[p] : 0.f;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (kbeg < kend) {
    load_slab(kbeg);
    store_slab(0);
    __syncthreads();
    int cur = 0;
    const int kh = lane >> 5, l31 = lane & 31;
    for (long long k0 = kbeg; k0 < kend; k0 += WBK) {
      const bool more = k0 + WBK < kend;
      if (more) load_slab(k0 + WBK);
#pragma unroll
      for (int kk = 0; kk < WBK / 2; ++kk) {
        float da[2], xb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) da[i] = sD[cur][kk * 2 + kh][(wco * 2 + i) * 32 + l31];
#pragma unroll
        for (int j = 0; j < 2; ++j) xb[j] = sXg[cur][kk * 2 + kh][(wj * 2 + j) * 32 + l31];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(da[i], xb[j], acc[i][j], 0, 0, 0);
      }
      if (more) store_slab(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }
  // D[i = co][j]: lane -> j column (lane & 31), reg r -> co row
  float* dwg = a.part ? a.part + ((size_t)blockIdx.y * a.groups + g) * a.cout_g * a.jtot
                      : a.dw + (size_t)g * a.cout_g * a.jtot;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int jj = j0 + (wj * 2 + j) * 32 + (lane & 31);
    if (jj >= a.jtot) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + (wco * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co >= a.cout_g) continue;
        float* d = dwg + (size_t)co * a.jtot + jj;
        if (a.part) *d = acc[i][j][r];
        else *d = (a.accumulate ? *d : 0.f) + acc[i][j][r] * a.scale;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------------
// Split-precision weight gradient: D[co][j = (ci,tap)] = sum_pix dy[co,pix] * x[ci, pix*s + tap - p] on the
// bf16 matrix pipe.  The reduction index (pixels) is the contiguous axis of BOTH global operands, which
// is exactly the k-contiguous [row][k] LDS layout the bf16 fragments want: a thread loads 4 consecutive
// pixels of one row (16 B; 8 lanes cover a 128 B line), splits them into limbs and writes 8 B per limb.
// K (= N*OH*OW) is split across blockIdx.y; partial tiles go to scratch and are summed in split order.
// ------------------------------------------------------------------------------------------------
template <int KS, int LIMBS>
__global__ __launch_bounds__(256) void conv_wgrad_split_kernel(const WgradArgs a) {
  constexpr int KK = KS * KS, MI = 2, NJ = 2;
  __shared__ __attribute__((aligned(16))) unsigned char sD[LIMBS][WT * ROWB];   // dy  [co][pix]
  __shared__ __attribute__((aligned(16))) unsigned char sXg[LIMBS][WT * ROWB];  // x   [j][pix]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wco = wid >> 1, wj = wid & 1;
  const unsigned ntiles = (unsigned)a.tiles_co * a.tiles_j;
  const unsigned logical = gg::xcd_remap(blockIdx.x, ntiles);
  const int tile_co = logical % a.tiles_co, tile_j = logical / a.tiles_co;
  const int g = blockIdx.z;
  const int co0 = tile_co * WT, j0 = tile_j * WT;
  const long long kbeg = (long long)blockIdx.y * a.k_per_split;
  long long kend = kbeg + a.k_per_split;
  if (kend > a.ktot) kend = a.ktot;
  const int ohw = a.oh * a.ow, hw = a.h * a.w;

  const int grp = tid & 7;          // 4-pixel group inside the 32-pixel slab
  const int r0 = tid >> 3;          // rows r0 + 32*q, q < 4
  int jci[4], jdy[4], jdx[4];
  bool cok[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int j = j0 + r0 + 32 * q;
    if (j < a.jtot) {
      const int ci = j / KK, tap = j - ci * KK;
      jci[q] = ci; jdy[q] = tap / KS - a.pad; jdx[q] = tap % KS - a.pad;
    } else {
      jci[q] = -1; jdy[q] = 0; jdx[q] = 0;
    }
    cok[q] = (co0 + r0 + 32 * q) < a.cout_g;
  }

  float4 rd[4];
  float rx[4][4];
  unsigned mx = 0;                 // validity bits of rx
  // (image, position inside the image) of this thread's first pixel, advanced by BKS per slab: the slab loop used to
  // decompose the 64-bit pixel index with a 64-bit division per slab - more VALU work than the slab's conversions
  int kn;
  unsigned krem;
  {
    const long long k = kbeg + grp * 4;
    kn = (int)(k / ohw);
    krem = (unsigned)(k - (long long)kn * ohw);
  }
  auto load_slab = [&](long long k0) {
    const long long k = k0 + grp * 4;                 // first of this thread's 4 pixels (same image row)
    const bool ok = k < kend;
    int n = 0, oy = 0, ox = 0;
    if (ok) {
      n = kn;
      oy = (int)(krem / (unsigned)a.ow);
      ox = (int)(krem - (unsigned)oy * (unsigned)a.ow);
    }
    krem += BKS;
    while (krem >= (unsigned)ohw) { krem -= (unsigned)ohw; ++kn; }
    const float* dyp = a.dy + ((size_t)(n * a.groups + g) * a.cout_g) * ohw + (size_t)oy * a.ow + ox;
    const float* xp = a.x + ((size_t)(n * a.groups + g) * a.cin_g) * hw;
    mx = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = co0 + r0 + 32 * q;
      const bool okd = ok & cok[q];
      rd[q] = *reinterpret_cast<const float4*>(dyp + (okd ? (size_t)co * ohw : 0));
      if (!okd) rd[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int iy = oy * a.stride + jdy[q];
      const bool rowok = ok & (jci[q] >= 0) & ((unsigned)iy < (unsigned)a.h);
      const float* xr = xp + (rowok ? (size_t)jci[q] * hw + (size_t)iy * a.w : 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ix = (ox + e) * a.stride + jdx[q];
        const unsigned v = (unsigned)rowok & (unsigned)((unsigned)ix < (unsigned)a.w);
        rx[q][e] = xr[v ? ix : 0];
        mx |= v << (q * 4 + e);
      }
    }
  };
  auto store_slab = [&]() {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float d[4] = {rd[q].x, rd[q].y, rd[q].z, rd[q].w};
      float x[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = ((mx >> (q * 4 + e)) & 1u) ? rx[q][e] : 0.f;
#pragma unroll
      for (int l = 0; l < LIMBS; ++l) {
        const unsigned d0 = pack_bf16x2(d[0], d[1]), d1 = pack_bf16x2(d[2], d[3]);
        const unsigned x0 = pack_bf16x2(x[0], x[1]), x1 = pack_bf16x2(x[2], x[3]);
        if (l + 1 < LIMBS) {
          d[0] -= bf16_lo(d0); d[1] -= bf16_hi(d0); d[2] -= bf16_lo(d1); d[3] -= bf16_hi(d1);
          x[0] -= bf16_lo(x0); x[1] -= bf16_hi(x0); x[2] -= bf16_lo(x1); x[3] -= bf16_hi(x1);
        }
        *reinterpret_cast<uint2*>(&sD[l][(r0 + 32 * q) * ROWB + grp * 8]) = make_uint2(d0, d1);
        *reinterpret_cast<uint2*>(&sXg[l][(r0 + 32 * q) * ROWB + grp * 8]) = make_uint2(x0, x1);
      }
    }
  };

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (kbeg < kend) {
    const int kh = lane >> 5, l31 = lane & 31;
    load_slab(kbeg);
    for (long long k0 = kbeg; k0 < kend; k0 += BKS) {
      store_slab();
      __syncthreads();
      if (k0 + BKS < kend) load_slab(k0 + BKS);
#pragma unroll
      for (int ks = 0; ks < BKS / 16; ++ks) {
        bf16x8 fa[LIMBS][MI], fb[LIMBS][NJ];
#pragma unroll
        for (int l = 0; l < LIMBS; ++l) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
            fa[l][i] = *reinterpret_cast<const bf16x8*>(&sD[l][((wco * MI + i) * 32 + l31) * ROWB + ks * 32 + kh * 16]);
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            fb[l][j] = *reinterpret_cast<const bf16x8*>(&sXg[l][((wj * NJ + j) * 32 + l31) * ROWB + ks * 32 + kh * 16]);
        }
#pragma unroll
        for (int sum = LIMBS - 1; sum >= 0; --sum)
#pragma unroll
          for (int la = 0; la <= sum; ++la) {
            const int lb = sum - la;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
              for (int j = 0; j < NJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[la][i], fb[lb][j], acc[i][j], 0, 0, 0);
          }
      }
      __syncthreads();
    }
  }
  float* dwg = a.part ? a.part + ((size_t)blockIdx.y * a.groups + g) * a.cout_g * a.jtot
                      : a.dw + (size_t)g * a.cout_g * a.jtot;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int jj = j0 + (wj * NJ + j) * 32 + (lane & 31);
    if (jj >= a.jtot) continue;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + (wco * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co >= a.cout_g) continue;
        float* d = dwg + (size_t)co * a.jtot + jj;
        if (a.part) *d = acc[i][j][r];
        else *d = (a.accumulate ? *d : 0.f) + acc[i][j][r] * a.scale;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 weight gradient, row-streaming formulation (split precision).
// The generic kernel above re-gathers and re-converts every x element once per tap (j = (ci, tap) columns): ~40
// VALU instructions per MFMA.  Here a block owns a (TCO co) x (TCI ci) x 9-tap tile and walks DOWN a vertical
// strip of the image (32 pixels wide), one output row per slab:
//   * the dy row segment (TCO x 32 px) is staged as [co][px] (k = px contiguous -> MFMA A operand);
//   * x lives in LDS as a rolling 3-row window [ci][row slot][48 px] (8-px aligned halo on both sides); each new
//     slab converts ONE new x row, and every x element then feeds all 9 taps;
//   * the B fragment of tap (ky, kx) is 8 consecutive pixels of row slot (y+ky-1) starting at px+kx-1: kx = 1
//     is an aligned 16-byte read, kx = 0 / 2 are built from it and one neighbouring dword with v_alignbit.
// Each wave owns 32 co x 32 ci and keeps the nine 32x32 accumulators (one per tap): 54 MFMAs per slab.
// K-splits (image, strip, row block) go to the workspace [tile][split][tap][co][ci] and are summed in split order by
// wgrad_reduce_kernel into dw (torch layout [co][ci][3][3]).
// ------------------------------------------------------------------------------------------------
// SW = strip width: 32 (one image row segment per 32-pixel slab) or 16 (16-wide images: a slab is TWO full rows, the
// window holds 4 rows and the k-half of a fragment selects the row instead of the column)
template <int LIMBS, int TCO, int TCI, bool MASK = false, int SW = 32>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_rows_kernel(const WgradArgs a, int segs, int rblocks,
                                                                    int rows_per_block, int units_per_block,
                                                                    float* __restrict__ ws,
                                                                    float* __restrict__ dbws) {
  static_assert(TCO * TCI == 4096, "four waves of 32 co x 32 ci");
  constexpr int WCI = TCI / 32;                       // ci waves; co waves = 4 / WCI
  constexpr int DPARTS = 256 / TCO, DPX = 32 / DPARTS;        // dy: threads per row, pixels per thread
  constexpr int XPT = TCI / 32;                       // x (ci, row, 4-px group) items per thread
  constexpr int ROWS = 32 / SW, NSLOT = ROWS + 2;     // image rows per slab, rows in the rolling window
  constexpr int SUBS = SW / 4;                        // 4-pixel groups per row
  constexpr int WR_XROW = (8 + SW + 8) * 2;           // bytes per x row slot: 8 halo + SW + 8 halo bf16
  constexpr int WR_XS = NSLOT * WR_XROW + 16;         // bytes per ci (odd multiple of 16 -> conflict-free lane stride)
  __shared__ __attribute__((aligned(16))) unsigned char sD[LIMBS][TCO * ROWB];
  __shared__ __attribute__((aligned(16))) unsigned char sX[LIMBS][TCI * WR_XS];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wci = wid % WCI, wco = wid / WCI;
  const int tiles_ci = a.tiles_j;
  const unsigned ntiles = (unsigned)a.tiles_co * tiles_ci;
  const unsigned logical = gg::xcd_remap(blockIdx.x, ntiles);
  const int tile_co = logical % a.tiles_co, tile_ci = logical / a.tiles_co;
  const int co0 = tile_co * TCO, ci0 = tile_ci * TCI;
  const int g = blockIdx.z;
  // A block sums over `units_per_block` consecutive K-units; unit -> (image, strip, row block).  (Small images: one
  // unit is only a few slabs, so several are chained to amortise the tile's epilogue.)
  const int hw = a.h * a.w;
  const int total_units = a.batch * segs * rblocks;
  int n = 0, c0 = 0, y0 = 0, y1 = 0;
  const float* xn = nullptr;
  const float* dsrc = nullptr;
  const float* msrc = nullptr;

  // ---- dy mover: row drow, pixels dpart*DPX .. +DPX
  const int drow = tid / DPARTS, dpart = tid % DPARTS;
  const bool d_ok = (co0 + drow) < a.cout_g;
  const float mpos = a.mask_gain, mneg = a.mask_gain * a.mask_alpha;
  float bsum = 0.f;                               // this thread's share of the bias gradient (MASK)
  auto set_unit = [&](int unit) {
    int sp = unit;
    const int rb = sp % rblocks; sp /= rblocks;
    const int sg = sp % segs;
    n = sp / segs;
    c0 = sg * SW;
    y0 = rb * rows_per_block;
    y1 = y0 + rows_per_block;
    if (y1 > a.h) y1 = a.h;
    xn = a.x + ((size_t)(n * a.groups + g) * a.cin_g) * hw;
    const size_t doff = ((size_t)(n * a.groups + g) * a.cout_g + (d_ok ? co0 + drow : 0)) * hw + c0 + dpart * DPX;
    dsrc = a.dy + doff;
    msrc = MASK ? a.mask_ref + doff : nullptr;
  };
  // ---- x mover: items (ci, 4-pixel group)
  int xci[XPT], xsub[XPT], xrr[XPT];
  bool x_ok[XPT];
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    const int item = tid + 256 * i;
    xci[i] = item >> 3;
    xsub[i] = item & (SUBS - 1);
    xrr[i] = (item & 7) / SUBS;                     // row inside the slab (0 for SW = 32)
    x_ok[i] = (ci0 + xci[i]) < a.cin_g;
  }

  float4 rd[DPX / 4];
  float4 rm[MASK ? DPX / 4 : 1];
  float4 rx[XPT];
  float rh[XPT];                                   // halo pixel (left for the first group, right for the last)
  auto load_dy = [&](int y) {
    const bool ok = d_ok & (y < y1);
#pragma unroll
    for (int q = 0; q < DPX / 4; ++q) {
      rd[q] = ok ? *reinterpret_cast<const float4*>(dsrc + (size_t)y * a.w + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (MASK)
        rm[MASK ? q : 0] = ok ? *reinterpret_cast<const float4*>(msrc + (size_t)y * a.w + q * 4)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto load_x = [&](int rowbase) {                  // rows rowbase .. rowbase + ROWS - 1
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int row = rowbase + xrr[i];
      const bool ok = ((unsigned)row < (unsigned)a.h) & x_ok[i];
      const float* src = xn + (size_t)(ok ? ci0 + xci[i] : 0) * hw + (size_t)(ok ? row : 0) * a.w + c0;
      rx[i] = ok ? *reinterpret_cast<const float4*>(src + xsub[i] * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      const int hc = xsub[i] == 0 ? c0 - 1 : c0 + SW;
      const bool hok = ok & ((xsub[i] == 0) | (xsub[i] == SUBS - 1)) & ((unsigned)hc < (unsigned)a.w);
      rh[i] = hok ? src[hc - c0] : 0.f;
    }
  };
  auto store_dy = [&]() {
    float v[DPX];
#pragma unroll
    for (int q = 0; q < DPX / 4; ++q) { v[4 * q] = rd[q].x; v[4 * q + 1] = rd[q].y; v[4 * q + 2] = rd[q].z; v[4 * q + 3] = rd[q].w; }
    if (MASK) {
#pragma unroll
      for (int q = 0; q < DPX / 4; ++q) {
        const float4 m = rm[MASK ? q : 0];
        v[4 * q] *= m.x > 0.f ? mpos : mneg;
        v[4 * q + 1] *= m.y > 0.f ? mpos : mneg;
        v[4 * q + 2] *= m.z > 0.f ? mpos : mneg;
        v[4 * q + 3] *= m.w > 0.f ? mpos : mneg;
      }
#pragma unroll
      for (int i = 0; i < DPX; ++i) bsum += v[i];
    }
#pragma unroll
    for (int l = 0; l < LIMBS; ++l) {
      unsigned pk[DPX / 2];
#pragma unroll
      for (int j = 0; j < DPX / 2; ++j) {
        pk[j] = pack_bf16x2(v[2 * j], v[2 * j + 1]);
        if (l + 1 < LIMBS) { v[2 * j] -= bf16_lo(pk[j]); v[2 * j + 1] -= bf16_hi(pk[j]); }
      }
      U4* dst = reinterpret_cast<U4*>(&sD[l][drow * ROWB + dpart * DPX * 2]);
#pragma unroll
      for (int q = 0; q < DPX / 8; ++q) dst[q] = U4{pk[4 * q], pk[4 * q + 1], pk[4 * q + 2], pk[4 * q + 3]};
    }
  };
  auto store_x = [&](int rowbase) {
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int slot = (rowbase + xrr[i] + NSLOT) % NSLOT;
      float v[4] = {rx[i].x, rx[i].y, rx[i].z, rx[i].w};
      float hv = rh[i];
      unsigned char* base = nullptr;
#pragma unroll
      for (int l = 0; l < LIMBS; ++l) {
        base = &sX[l][xci[i] * WR_XS + slot * WR_XROW];
        const unsigned p0 = pack_bf16x2(v[0], v[1]), p1 = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>(base + (8 + xsub[i] * 4) * 2) = make_uint2(p0, p1);
        const __bf16 hb = (__bf16)hv;
        if (xsub[i] == 0) *reinterpret_cast<__bf16*>(base + 7 * 2) = hb;
        if (xsub[i] == SUBS - 1) *reinterpret_cast<__bf16*>(base + (8 + SW) * 2) = hb;
        if (l + 1 < LIMBS) {
          v[0] -= bf16_lo(p0); v[1] -= bf16_hi(p0); v[2] -= bf16_lo(p1); v[3] -= bf16_hi(p1);
          hv -= (float)hb;
        }
      }
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int kh = lane >> 5, l31 = lane & 31;
  const int a_off = (wco * 32 + l31) * ROWB + kh * 16;               // + ks*32
  const int b_off = (wci * 32 + l31) * WR_XS + (8 + kh * 8) * 2;     // + slot*WR_XROW (+ ks*32 for SW = 32)

  for (int u = 0; u < units_per_block; ++u) {
    const int unit = blockIdx.y * units_per_block + u;
    if (unit >= total_units) break;
    set_unit(unit);
    if (y0 >= y1) continue;
    // prologue: rows y0-1 and y0 of x
    for (int r = y0 - 1; r < y0 + 1; r += ROWS) {
      load_x(r);
      store_x(r);
    }
    load_dy(y0);
    load_x(y0 + 1);
    for (int y = y0; y < y1; y += ROWS) {
      store_dy();
      store_x(y + 1);
      __syncthreads();
      load_dy(y + ROWS);
      load_x(y + ROWS + 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 fa[LIMBS];
#pragma unroll
        for (int l = 0; l < LIMBS; ++l) fa[l] = *reinterpret_cast<const bf16x8*>(&sD[l][a_off + ks * 32]);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          // the slab's k-half ks is columns 16 ks.. of row y (SW = 32) or the whole row y + ks (SW = 16)
          const int slot = (y + (SW == 16 ? ks : 0) + ky - 1 + NSLOT) % NSLOT;
          bf16x8 fb[LIMBS][3];
#pragma unroll
          for (int l = 0; l < LIMBS; ++l) {
            const unsigned char* p = &sX[l][b_off + slot * WR_XROW + (SW == 32 ? ks * 32 : 0)];
            const U4 mid = *reinterpret_cast<const U4*>(p);
            const unsigned prev = *reinterpret_cast<const unsigned*>(p - 4);
            const unsigned next = *reinterpret_cast<const unsigned*>(p + 16);
            const U4 left{__builtin_amdgcn_alignbit(mid[0], prev, 16), __builtin_amdgcn_alignbit(mid[1], mid[0], 16),
                          __builtin_amdgcn_alignbit(mid[2], mid[1], 16), __builtin_amdgcn_alignbit(mid[3], mid[2], 16)};
            const U4 right{__builtin_amdgcn_alignbit(mid[1], mid[0], 16), __builtin_amdgcn_alignbit(mid[2], mid[1], 16),
                           __builtin_amdgcn_alignbit(mid[3], mid[2], 16), __builtin_amdgcn_alignbit(next, mid[3], 16)};
            fb[l][0] = __builtin_bit_cast(bf16x8, left);              // kx = 0: pixels px-1 ..
            fb[l][1] = __builtin_bit_cast(bf16x8, mid);               // kx = 1
            fb[l][2] = __builtin_bit_cast(bf16x8, right);             // kx = 2: pixels px+1 ..
          }
#pragma unroll
          for (int sum = LIMBS - 1; sum >= 0; --sum)
#pragma unroll
            for (int la = 0; la <= sum; ++la) {
              const int lb = sum - la;
#pragma unroll
              for (int kx = 0; kx < 3; ++kx)
                acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[la], fb[lb][kx], acc[ky * 3 + kx], 0, 0, 0);
            }
        }
      }
      __syncthreads();
    }
  }

  if (MASK && dbws && tile_ci == 0) {
    // bias gradient: per-row sums of this block, one plain store per (split, channel); summed by wgrad_reduce_kernel
    float* sdb = reinterpret_cast<float*>(&sD[0][0]);          // the main loop's last barrier freed sD
    float rowsum = bsum;                                       // the DPARTS movers of a row are neighbouring lanes
#pragma unroll
    for (int m = 1; m < DPARTS; m <<= 1) rowsum += __shfl_xor(rowsum, m, 64);
    if (dpart == 0) sdb[drow] = rowsum;
    __syncthreads();
    if (tid < TCO && co0 + tid < a.cout_g)
      dbws[((size_t)blockIdx.y * a.groups + g) * a.cout_g + co0 + tid] = sdb[tid];
  }
  if (ws) {
    // partial tile -> workspace [tile][split][tap][co][ci] (ci fastest: 128-byte runs per store); summed by
    // wgrad_reduce_kernel.  (Atomics straight into dw serialise: blocks x 36,864 adds onto few addresses.)
    const size_t tile = ((size_t)g * a.tiles_co + tile_co) * tiles_ci + tile_ci;
    float* base = ws + (tile * gridDim.y + blockIdx.y) * (size_t)(9 * TCO * TCI);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int col = wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        base[((size_t)t * TCO + col) * TCI + wci * 32 + l31] = acc[t][r];
      }
    return;
  }
  // (no workspace: not launched - the host falls back to the generic kernel)
}

// 1x1 / stride 1 weight gradient with very few input channels (the STN's 3 -> 64 RGB stem): dW[co][ci] =
// sum_pixels dy[co][p] * x[ci][p] is a bandwidth-bound reduction of dy (67 MB at 128^2, batch 16) - the GEMM kernels
// spent 0.17 ms on it with 3 of their 128 columns in use.  Lanes run along pixels, a wave owns 16 output channels and
// keeps the 16 x CIN partial sums in registers; per-block partials go to the workspace and are summed by
// partial_sum_kernel.
template <int CIN>
__global__ __launch_bounds__(256) void wgrad_1x1_smallcin_kernel(float* __restrict__ ws, const float* __restrict__ x,
                                                                 const float* __restrict__ dy, int batch, int cout,
                                                                 long long hw) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int co0 = blockIdx.y * 64 + wid * 16;
  float acc[16][CIN];
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int c = 0; c < CIN; ++c) acc[r][c] = 0.f;
  const long long chunks_per_img = hw / 64, chunks = (long long)batch * chunks_per_img;
  for (long long ch = blockIdx.x; ch < chunks; ch += gridDim.x) {
    const long long n = ch / chunks_per_img, p = (ch - n * chunks_per_img) * 64 + lane;
    float xv[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) xv[c] = x[((size_t)n * CIN + c) * hw + p];
    const float* dyn = dy + ((size_t)n * cout) * hw + p;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d = (co0 + r < cout) ? dyn[(size_t)(co0 + r) * hw] : 0.f;
#pragma unroll
      for (int c = 0; c < CIN; ++c) acc[r][c] += d * xv[c];
    }
  }
  float* dst = ws + (size_t)blockIdx.x * cout * CIN;
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
      const float t = gg::wave_sum(acc[r][c]);
      if (lane == 0 && co0 + r < cout) dst[(co0 + r) * CIN + c] = t;
    }
}

// 3x3 / stride 1 / pad 1 weight gradient with <= 4 OUTPUT channels (the flow head's last layer, 512 -> 2 @16^2: the generic
// 128 x 128-tile kernel spent 78 us on it with 2 of its 128 rows in use).  dW[co][ci][tap] = sum_{n,p} dy[n,co,p] x[n,ci,p+tap]:
// one block per input channel, threads stride over (sample, pixel) with the NCO x 9 sums in registers, then the fixed tree
// of block_sum_256 per output - no partial tiles, no reduce pass, exact fp32 products, reproducible.
template <int NCO>
__global__ __launch_bounds__(256) void wgrad3x3_fewout_kernel(float* __restrict__ dw, const float* __restrict__ x,
                                                              const float* __restrict__ dy, int batch, int cin, int h,
                                                              int w, float scale, int accumulate) {
  __shared__ float red[4];
  const int ci = blockIdx.x;
  const unsigned hw = (unsigned)h * (unsigned)w, total = (unsigned)batch * hw;
  float acc[NCO][9];
#pragma unroll
  for (int j = 0; j < NCO; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[j][t] = 0.f;
  for (unsigned i = threadIdx.x; i < total; i += 256u) {
    const unsigned n = i / hw, p = i - n * hw;
    const int py = (int)(p / (unsigned)w), px = (int)(p - (unsigned)py * (unsigned)w);
    float d[NCO];
#pragma unroll
    for (int j = 0; j < NCO; ++j) d[j] = dy[((size_t)n * NCO + j) * hw + p];
    const float* xp = x + ((size_t)n * cin + ci) * hw;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = py + ky - 1, ix = px + kx - 1;
        const bool ok = (unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w;
        const float v = ok ? xp[iy * w + ix] : 0.f;
#pragma unroll
        for (int j = 0; j < NCO; ++j) acc[j][ky * 3 + kx] = fmaf(d[j], v, acc[j][ky * 3 + kx]);
      }
  }
#pragma unroll
  for (int j = 0; j < NCO; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float tot = gg::block_sum_256<float>(acc[j][t], red);
      if (threadIdx.x == 0) {
        float* dst = dw + ((size_t)j * cin + ci) * 9 + t;
        *dst = fmaf(tot, scale, accumulate ? *dst : 0.f);
      }
    }
}

// Weight gradient of a convolution whose OUTPUT is tiny (<= 32 positions per image: the similarity trunk's layers at 4^2 -
// the 1x1 skip; the 3x3 layers there were tried and stay on the generic tile, see wgrad_entry).  OH * OW is not a multiple of 32 there, so the split-precision
// K-slab loaders do not apply and the launches fell to the exact-fp32 generic tile: K = 256 pixels in all, 78 us for 1.2
// GFLOP.  Here a block owns 32 output x 8 input channels and walks the samples: dy[n, 32 co, P] and x[n, 8 ci, H * W] are
// staged in LDS - x in im2col form [ci][position][tap], gathered once per block and sample through a table of the (position,
// tap) offsets - and a thread (co, ci) adds its KK taps from contiguous rows.  Exact fp32 products, samples and positions in
// ascending order.
constexpr int TS_CO = 32, TS_CI = 8, TS_MAXP = 32, TS_MAXHW = 256;
template <int KS>
__global__ __launch_bounds__(256) void wgrad_tiny_spatial_kernel(float* __restrict__ dw, const float* __restrict__ x,
                                                                 const float* __restrict__ dy, int batch, int cin,
                                                                 int cout, int h, int w, int oh, int ow, int stride,
                                                                 int pad, float scale, int accumulate) {
  constexpr int KK = KS * KS;
  __shared__ float sdy[TS_CO][TS_MAXP + 1];
  __shared__ float sxc[TS_CI][TS_MAXP * KK + 1];            // x of one sample in im2col form: [ci][position][tap]
  __shared__ short sidx[TS_MAXP * KK];                      // x offset of (position, tap); -1 = padding
  const int tid = threadIdx.x, co_l = tid & (TS_CO - 1), ci_l = tid >> 5;
  const int co0 = blockIdx.x * TS_CO, ci0 = blockIdx.y * TS_CI;
  const int P = oh * ow, hw = h * w, PK = P * KK;
  for (int i = tid; i < PK; i += 256) {
    const int p = i / KK, t = i - p * KK;
    const int iy = (p / ow) * stride + t / KS - pad, ix = (p % ow) * stride + t % KS - pad;
    sidx[i] = ((unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w) ? (short)(iy * w + ix) : (short)-1;
  }
  float acc[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t) acc[t] = 0.f;
  for (int n = 0; n < batch; ++n) {
    __syncthreads();                                        // (also publishes sidx before its first use)
    for (int i = tid; i < TS_CO * P; i += 256) {
      const int r = i / P, p = i - r * P;
      sdy[r][p] = (co0 + r < cout) ? dy[((size_t)n * cout + co0 + r) * P + p] : 0.f;
    }
    for (int i = tid; i < TS_CI * PK; i += 256) {           // the gather happens once per block and sample, not per thread
      const int r = i / PK, e = i - r * PK;
      const int off = sidx[e];
      sxc[r][e] = (off >= 0 && ci0 + r < cin) ? x[((size_t)n * cin + ci0 + r) * hw + off] : 0.f;
    }
    __syncthreads();
#pragma unroll 2
    for (int p = 0; p < P; ++p) {
      const float d = sdy[co_l][p];
      const float* row = &sxc[ci_l][p * KK];
#pragma unroll
      for (int t = 0; t < KK; ++t) acc[t] = fmaf(d, row[t], acc[t]);
    }
  }
  const int co = co0 + co_l, ci = ci0 + ci_l;
  if (co < cout && ci < cin) {
    float* dst = dw + ((size_t)co * cin + ci) * KK;
#pragma unroll
    for (int t = 0; t < KK; ++t) dst[t] = fmaf(acc[t], scale, accumulate ? dst[t] : 0.f);
  }
}

// out[i] (+)= scale * sum_b ws[b * count + i]: one wave per output element, lanes stride over the partials
__global__ __launch_bounds__(256) void partial_sum_kernel(float* __restrict__ out, const float* __restrict__ ws,
                                                          int nblocks, int count, float scale, int accumulate) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= count) return;
  float sum = 0.f;
  for (int b = lane; b < nblocks; b += 64) sum += ws[(size_t)b * count + i];
  sum = gg::wave_sum(sum);
  if (lane == 0) out[i] = (accumulate ? out[i] : 0.f) + sum * scale;
}

// out[i] (+)= scale * sum_b ws[b * count + i], one thread per element (few partials, many elements; coalesced)
__global__ __launch_bounds__(256) void partial_sum_flat_kernel(float* __restrict__ out, const float* __restrict__ ws,
                                                               int nblocks, long long count, float scale,
                                                               int accumulate) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
    float sum = ws[i];
    for (int b = 1; b < nblocks; ++b) sum += ws[(size_t)b * count + i];
    out[i] = (accumulate ? out[i] : 0.f) + sum * scale;
  }
}

// dw[g][co][ci][tap] (+)= scale * sum_split ws[tile][split][tap][co_l][ci_l]
// RG lanes share one output element and each sums every RG-th split (then a shuffle reduction): with one thread per
// element the narrow layers (one tile, 512 splits) were a 512-long chain of dependent loads on half a block per CU.
// RG = 1 (one element per lane, fully coalesced) serves the many-tile / few-split layers.
template <int RG>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(float* __restrict__ dw, const float* __restrict__ ws,
                                                           int groups, int cout_g, int cin_g, int tiles_co,
                                                           int tiles_ci, int tco, int tci, int splits, float scale,
                                                           int accumulate, float* __restrict__ dbias,
                                                           const float* __restrict__ dbws) {
  const size_t tile_elems = (size_t)9 * tco * tci;
  const size_t total = (size_t)groups * tiles_co * tiles_ci * tile_elems;
  if (dbias) {        // bias gradient: dbias[c] += sum over splits of the per-block row sums
    const int channels = groups * cout_g;
    for (size_t c = (size_t)blockIdx.x * 256 + threadIdx.x; c < (size_t)channels; c += (size_t)gridDim.x * 256) {
      float sum = 0.f;
      for (int sidx = 0; sidx < splits; ++sidx) sum += dbws[(size_t)sidx * channels + c];
      dbias[c] += sum;
    }
  }
  // thread t: element (t / 64) * (64 / RG) ... : lanes [0, 64/RG) of each RG-row are consecutive elements
  const int lane = threadIdx.x & 63, sub = lane / (64 / RG), el = lane % (64 / RG);
  const size_t waves = (size_t)gridDim.x * 4;
  for (size_t base = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / RG); base < total; base += waves * (64 / RG)) {
    const size_t o = base + el;
    float sum = 0.f;
    size_t tile = 0;
    int e = 0;
    const bool ok = o < total;
    if (ok) {
      tile = o / tile_elems;
      e = (int)(o - tile * tile_elems);
      const float* src = ws + tile * splits * tile_elems + e;
      for (int sidx = sub; sidx < splits; sidx += RG) sum += src[(size_t)sidx * tile_elems];
    }
#pragma unroll
    for (int m = 64 / RG; m < 64; m <<= 1) sum += __shfl_xor(sum, m, 64);
    if (!ok || sub != 0) continue;
    const int ci_l = e % tci, co_l = (e / tci) % tco, t = e / (tci * tco);
    const int tile_ci = (int)(tile % tiles_ci), tile_co = (int)((tile / tiles_ci) % tiles_co);
    const int g = (int)(tile / ((size_t)tiles_ci * tiles_co));
    const int co = tile_co * tco + co_l, ci = tile_ci * tci + ci_l;
    if (co >= cout_g || ci >= cin_g) continue;
    float* dst = dw + (((size_t)g * cout_g + co) * cin_g + ci) * 9 + t;
    *dst = (accumulate ? *dst : 0.f) + sum * scale;
  }
}

// wmat[g][(r, ky, kx)][c], r = reduction channel (cin_g of them), c = output channel (cout_g)
__device__ __forceinline__ void pack_weight_body(float* __restrict__ wmat, const float* __restrict__ w, long long total,
                                                 int cout_g, int cin_g, int kh, int kw, int transpose_io, int flip,
                                                 float scale, long long first, long long stride) {
  const int kk = kh * kw;
  for (long long o = first; o < total; o += stride) {
    const int c = (int)(o % cout_g);
    long long q = o / cout_g;
    const int tap = (int)(q % kk);
    q /= kk;
    const int r = (int)(q % cin_g);
    const int g = (int)(q / cin_g);
    int ky = tap / kw, kx = tap % kw;
    if (flip) { ky = kh - 1 - ky; kx = kw - 1 - kx; }
    const size_t src = transpose_io ? (((size_t)g * cin_g + r) * cout_g + c) * kk + ky * kw + kx
                                    : (((size_t)g * cout_g + c) * cin_g + r) * kk + ky * kw + kx;
    wmat[o] = w[src] * scale;
  }
}

__global__ __launch_bounds__(256) void pack_weight_kernel(float* __restrict__ wmat, const float* __restrict__ w,
                                                          long long total, int cout_g, int cin_g, int kh, int kw,
                                                          int transpose_io, int flip, float scale) {
  pack_weight_body(wmat, w, total, cout_g, cin_g, kh, kw, transpose_io, flip, scale,
                   (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x);
}

// per-plane dot product: one block per plane, 16 B loads when possible
__global__ __launch_bounds__(256) void plane_dot_kernel(float* __restrict__ out, const float* __restrict__ a,
                                                        const float* __restrict__ b, long long hw) {
  __shared__ float red[4];
  const size_t base = (size_t)blockIdx.x * hw;
  float acc = 0.f;
  if ((hw & 3) == 0) {
    const float4* a4 = reinterpret_cast<const float4*>(a + base);
    const float4* b4 = reinterpret_cast<const float4*>(b + base);
    for (long long i = threadIdx.x; i < hw / 4; i += 256) {
      const float4 u = a4[i], v = b4[i];
      acc += u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w;
    }
  } else {
    for (long long i = threadIdx.x; i < hw; i += 256) acc += a[base + i] * b[base + i];
  }
  const float tot = gg::block_sum_256<float>(acc, red);
  if (threadIdx.x == 0) out[blockIdx.x] = tot;
}

// y = epilogue(sum over splits of part[s]), splits added in ascending order (see ConvArgs::part).  The epilogue is the
// one the un-split kernels apply: * out_scale[n, c], + bias[c], then optionally the StyledConv tail
// lrelu(v + noise_w * noise[n, pix] + act_bias[c]) * gain.  VEC = 4 needs ohw % 4 == 0 (a float4 never straddles planes).
template <int VEC>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(float* __restrict__ y, const float* __restrict__ part,
                                                            long long stride, int splits, long long nvec,
                                                            const ConvArgs a) {
  const long long ohw = (long long)a.oh * a.ow;
  const int C = a.groups * a.cout_g;
  const float anw = (a.act && a.act_noise) ? a.act_noise_w[0] : 0.f;
  const long long gstride = (long long)gridDim.x * blockDim.x;
  const bool small = nvec * VEC < (1LL << 31);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += gstride) {
    const long long e = i * VEC;
    float v[VEC];
    if (VEC == 4) {
      float4 t = *reinterpret_cast<const float4*>(part + e);
      for (int sidx = 1; sidx < splits; ++sidx) {
        const float4 u = *reinterpret_cast<const float4*>(part + (size_t)sidx * stride + e);
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      v[0] = t.x; v[1 % VEC] = t.y; v[2 % VEC] = t.z; v[3 % VEC] = t.w;
    } else {
      float t = part[e];
      for (int sidx = 1; sidx < splits; ++sidx) t += part[(size_t)sidx * stride + e];
      v[0] = t;
    }
    // (32-bit index math when the tensor allows it: each 64-bit division is ~100 VALU instructions)
    long long plane, n, p;
    int c;
    if (small) {
      const unsigned pl = (unsigned)e / (unsigned)ohw, nn = pl / (unsigned)C;
      plane = pl; n = nn; c = (int)(pl - nn * (unsigned)C); p = (unsigned)e - pl * (unsigned)ohw;
    } else {
      plane = e / ohw; n = plane / C; c = (int)(plane - n * C); p = e - plane * ohw;
    }
    const float sc = a.acc_scale * (a.out_scale ? a.out_scale[plane] : 1.f), bi = a.bias ? a.bias[c] : 0.f;
#pragma unroll
    for (int q = 0; q < VEC; ++q) v[q] = v[q] * sc + bi;
    if (a.act) {
      const float ab = a.act_bias ? a.act_bias[c] : 0.f;
#pragma unroll
      for (int q = 0; q < VEC; ++q) {
        const float t = v[q] + (a.act_noise ? anw * a.act_noise[n * ohw + p + q] : 0.f) + ab;
        v[q] = (t > 0.f ? t : t * a.act_alpha) * a.act_gain;
      }
    }
    if (a.residual) {
#pragma unroll
      for (int q = 0; q < VEC; ++q) v[q] += a.residual[e + q];
    }
    if (VEC == 4) *reinterpret_cast<float4*>(y + e) = make_float4(v[0], v[1 % VEC], v[2 % VEC], v[3 % VEC]);
    else y[e] = v[0];
  }
}

// Which kernel instantiation the last convolution entry point of this thread launched (gg_last_conv_kernel): what
// bench.py's per-kernel timing keys on, so that its roofline entries name the kernel that actually ran.
thread_local const unsigned short* g_prelimb = nullptr;      // gg_debug_set_prelimb (measurement only)
thread_local int g_prelimb_served = 0;
thread_local char g_last_kernel[96] = "";
thread_local int g_sign_bits_written = 0;      // the last forward launch wrote the sign plane it was handed (BitArgs)
thread_local int g_amax_written = 0;           // ... and the per-image output maxima (BitArgs::amax)
#define NOTE_KERNEL(...) snprintf(g_last_kernel, sizeof(g_last_kernel), __VA_ARGS__)

// Scratch for `splits` partial copies of the output of `a`; sets a.part / a.part_stride.  zero: clear it first
// (launches that do not cover every output element: parity-class plans of the generic transposed path).
int splitk_prepare(ConvArgs& a, int splits, bool zero, hipStream_t st) {
  const long long elems = (long long)a.batch * a.groups * a.cout_g * a.oh * a.ow;
  a.part_stride = (elems + 3) / 4 * 4;
  const size_t bytes = sizeof(float) * (size_t)a.part_stride * splits;
  a.part = reinterpret_cast<float*>(gg::scratch(st, bytes));
  if (!a.part) return -3;
  if (zero) {
    hipError_t e = hipMemsetAsync(a.part, 0, bytes, st);
    if (e != hipSuccess) return gg::fail((int)e, "conv2d: memset failed");
  }
  return 0;
}

int splitk_reduce(const ConvArgs& a, int splits, hipStream_t st) {
  const long long elems = (long long)a.batch * a.groups * a.cout_g * a.oh * a.ow;
  const long long ohw = (long long)a.oh * a.ow;
  if (ohw % 4 == 0 && (reinterpret_cast<uintptr_t>(a.y) & 15) == 0 &&
      (!a.act_noise || (reinterpret_cast<uintptr_t>(a.act_noise) & 3) == 0))
    splitk_reduce_kernel<4><<<gg::stream_grid(elems / 4, 256), 256, 0, st>>>(a.y, a.part, a.part_stride, splits,
                                                                             elems / 4, a);
  else
    splitk_reduce_kernel<1><<<gg::stream_grid(elems, 256), 256, 0, st>>>(a.y, a.part, a.part_stride, splits, elems, a);
  return gg::launch_status("splitk_reduce");
}

// Fill in tiling / split-K for one launch.  Returns false when the launch is empty.
// tile: 0 = 128co x 128pix, 1 = 32co x 256pix, 2 = 64co x 256pix, 4 = 128co x 256pix (split kernels)
bool plan_conv(ConvArgs& a, int tile, int bk = BK) {
  const int tco = (tile == 0 || tile == 4) ? 128 : (tile == 1 ? 32 : 64), tpix = tile == 0 ? 128 : 256;
  a.tile_pixels = tpix;
  const long long mtot = (long long)a.batch * a.mh * a.mw;
  if (mtot <= 0) return false;
  a.tiles_co = (a.cout_g + tco - 1) / tco;
  const long long tp = (mtot + tpix - 1) / tpix;
  a.tiles_pix = (int)tp;
  a.nslabs = (a.ktot + bk - 1) / bk;
  const long long blocks = tp * a.tiles_co * a.groups;
  int splitk = 1;
  if (blocks < 2 * gg::kNumCu) {
    splitk = (int)((2 * gg::kNumCu + blocks - 1) / blocks);
    const int max_split = (a.nslabs * bk + 127) / 128;      // keep >= 128 k per split
    if (splitk > max_split) splitk = max_split;
    if (splitk < 1) splitk = 1;
  }
  a.slabs_per_split = (a.nslabs + splitk - 1) / splitk;
  splitk = (a.nslabs + a.slabs_per_split - 1) / a.slabs_per_split;
  a.splitk = splitk < 1 ? 1 : splitk;
  return true;
}

template <int KS, int MODE>
int launch_conv(const ConvArgs& a, int tile, hipStream_t st) {
  if ((long long)a.tiles_pix * a.tiles_co >= (1LL << 31)) return gg::fail(-2, "conv2d: too many tiles");
  dim3 grid((unsigned)(a.tiles_pix * a.tiles_co), (unsigned)a.splitk, (unsigned)a.groups);
  const bool sc = a.in_scale != nullptr;
  NOTE_KERNEL("conv_igemm<k%d,mode%d,tile%d,fp32>", KS, MODE, tile);
  if (tile == 1) {
    if (sc) conv_igemm_kernel<KS, MODE, 1, 4, 1, 2, true><<<grid, 256, 0, st>>>(a);
    else conv_igemm_kernel<KS, MODE, 1, 4, 1, 2, false><<<grid, 256, 0, st>>>(a);
  } else if (tile == 2) {
    if (sc) conv_igemm_kernel<KS, MODE, 1, 4, 2, 2, true><<<grid, 256, 0, st>>>(a);
    else conv_igemm_kernel<KS, MODE, 1, 4, 2, 2, false><<<grid, 256, 0, st>>>(a);
  } else {
    if (sc) conv_igemm_kernel<KS, MODE, 2, 2, 2, 2, true><<<grid, 256, 0, st>>>(a);
    else conv_igemm_kernel<KS, MODE, 2, 2, 2, 2, false><<<grid, 256, 0, st>>>(a);
  }
  return gg::launch_status("conv_igemm");
}

// One or two bf16 limbs per operand share the tile choices ("bf16": a single product, |error| ~ 2^-9 |a||b| - the
// arithmetic BASELINE.json's benchmark configuration names; "bf16x3": see above).  LIMBS12 instantiates a launch for both.
#define LIMBS12(limbs, ...)            \
  do {                                 \
    if ((limbs) == 1) {                \
      constexpr int L = 1;             \
      __VA_ARGS__;                     \
    } else {                           \
      constexpr int L = 2;             \
      __VA_ARGS__;                     \
    }                                  \
  } while (0)

// 256-pixel tiles (8 waves) halve the weight stream per output; use them while they still give >= 2 blocks per CU
int split_tile(const ConvArgs& a, int limbs) {
  const long long tiles256 = ((long long)a.batch * a.mh * a.mw + 255) / 256 * ((a.cout_g + 127) / 128) * a.groups;
  // (A patch-reuse kernel was costed for stride 2 and dropped: a 128-pixel output tile needs a (2TH+1) x (2TW+1) input
  // patch = 4.5 input pixels per output, which only fits LDS with 16-channel chunks, i.e. 12 MFMAs per barrier
  // interval - no better than this kernel's 24 with its per-tap gathers.)
  // stride-2 correlations re-gather per tap: two unsynchronised 4-wave blocks per CU overlap gather and MFMA phases
  // better than one 8-wave block (measured 175 -> 205 TF/s on the generator's up-conv data gradients)
  if (a.bs == 2) return 0;
  return (limbs <= 2 && tiles256 >= 2 * gg::kNumCu) ? 4 : 0;
}

template <int KS, int MODE>
int launch_conv_split(const ConvArgs& a, int limbs, hipStream_t st) {
  if ((long long)a.tiles_pix * a.tiles_co >= (1LL << 31)) return gg::fail(-2, "conv2d: too many tiles");
  dim3 grid((unsigned)(a.tiles_pix * a.tiles_co), (unsigned)a.splitk, (unsigned)a.groups);
  const bool sc = a.in_scale != nullptr;
  NOTE_KERNEL("conv_split<k%d,mode%d,s%d,limbs%d,%dpx,%s>", KS, MODE, a.bs, limbs, a.tile_pixels, a.f16 ? "f16" : "bf16");
  if (a.f16) {          // two binary16 limbs (forward convolutions of the fp16x3 mode)
    if (a.tile_pixels == 256) {
      if (sc) conv_split_kernel<KS, MODE, 2, true, 256, true><<<grid, 512, 0, st>>>(a);
      else conv_split_kernel<KS, MODE, 2, false, 256, true><<<grid, 512, 0, st>>>(a);
    } else {
      if (sc) conv_split_kernel<KS, MODE, 2, true, 128, true><<<grid, 256, 0, st>>>(a);
      else conv_split_kernel<KS, MODE, 2, false, 128, true><<<grid, 256, 0, st>>>(a);
    }
  } else if (limbs <= 2 && a.tile_pixels == 256) {
    if (sc) LIMBS12(limbs, conv_split_kernel<KS, MODE, L, true, 256><<<grid, 512, 0, st>>>(a));
    else LIMBS12(limbs, conv_split_kernel<KS, MODE, L, false, 256><<<grid, 512, 0, st>>>(a));
  } else if (limbs <= 2) {
    if (sc) LIMBS12(limbs, conv_split_kernel<KS, MODE, L, true, 128><<<grid, 256, 0, st>>>(a));
    else LIMBS12(limbs, conv_split_kernel<KS, MODE, L, false, 128><<<grid, 256, 0, st>>>(a));
  } else {
    if (sc) conv_split_kernel<KS, MODE, 3, true, 128><<<grid, 256, 0, st>>>(a);
    else conv_split_kernel<KS, MODE, 3, false, 128><<<grid, 256, 0, st>>>(a);
  }
  return gg::launch_status("conv_split");
}

// the StyledConv tail as a separate in-place pass (launches whose epilogue cannot carry it)
int post_activation(const ConvArgs& a, hipStream_t st) {
  if (!a.act_noise)
    return gg_fused_bias_act_f32(a.y, a.y, a.act_bias, nullptr, 3, 0, a.act_alpha, a.act_gain,
                                 (long long)a.batch * a.groups * a.cout_g * a.oh * a.ow, (long long)a.oh * a.ow,
                                 a.act_bias ? a.groups * a.cout_g : 0, st);
  return gg_noise_bias_act_f32(a.y, a.y, a.act_noise, a.act_noise_w, a.act_bias, a.act_alpha, a.act_gain, a.batch,
                               a.groups * a.cout_g, (long long)a.oh * a.ow, st);
}

// split-K threshold of the patch kernels: a launch with fewer blocks than this is split along Cin until it has about
// that many.  (GG_SPLIT_PATCH / GG_SPLIT_CONVT: measurement overrides.)
static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
// 256-pixel (8-wave, one block per CU) tiles halve the weight stream per output, which pays once the reduction is
// deep; with <= 64 input channels a tile's main loop is two chunks long and two co-resident 128-pixel blocks hide
// each other's prologue / epilogue instead (64 -> 64 @128^2, batch 16: 80 vs 69 us)
// switches that tests and A/B sessions flip at run time (gg_set_tuning); initialised from the environment
enum { kTunConvT16 = 0, kTunConvT16Tw, kTunCount };
struct Tunable {
  const char* name;
  int dflt, value;
  bool init;
};
static Tunable g_tunables[kTunCount] = {{"GG_CONVT16", 1, 0, false}, {"GG_CONVT16_TW", 64, 0, false}};
static int tuning(int idx) {
  Tunable& t = g_tunables[idx];
  if (!t.init) {
    t.value = env_int(t.name, t.dflt);
    t.init = true;
  }
  return t.value;
}
static int patch256_min_cin() {
  static const int v = env_int("GG_PATCH256_MIN_CIN", 64);
  return v;
}
static int split_at_patch() {
  static const int v = env_int("GG_SPLIT_PATCH", 2 * gg::kNumCu);
  return v;
}
// The transposed kernel's split-K epilogue is four classes of scattered 4-byte stores plus a reduce pass, so it only
// pays when most of the chip would otherwise idle (thresholds measured with the round-2 atomic epilogue).  Measured per generator pass at batch 16 (512 -> 512 channels):
// 16^2 -> 33^2 (384 blocks) 0.308 ms split in two vs 0.143 unsplit; 8^2 -> 17^2 (192 blocks) 0.154 vs 0.124;
// 4^2 -> 9^2 (128 blocks) 0.093 split vs 0.121 unsplit.
static int split_at_convt() {
  static const int v = env_int("GG_SPLIT_CONVT", 3 * gg::kNumCu / 4);
  return v;
}

// 3x3 / stride 1 / pad 1 with a power-of-two width >= 16 whose 128-pixel tiles fit the image
bool patch_geometry(const ConvArgs& a, int tpix, int& tw_log2, int limbs = 2) {
  const int w = a.w, h = a.h;
  if (w < 16 || (w & (w - 1)) != 0) return false;
  if ((long long)a.cin_g * h * w * 4 >= (1LL << 31)) return false;        // buffer-resource addressing
  static const int tw_env = env_int("GG_PATCH_TW", 64);      // measurement override: narrower, taller tiles
  const int tw_max = limbs == 3 ? 32 : tw_env;
  int tw = w < tw_max ? w : tw_max;
  tw_log2 = 0;
  while ((1 << tw_log2) < tw) ++tw_log2;
  const int th = tpix >> tw_log2;
  return th <= h && h % th == 0;
}

int launch_conv_patch(ConvArgs a, int limbs, int tw_log2, int tpix, hipStream_t st) {
  // plan in units of 32-channel chunks
  a.mh = a.oh; a.mw = a.ow;
  const bool narrow = limbs <= 2 && a.cout_g <= 64;           // 64-channel tiles
  a.tiles_co = narrow ? 1 : (a.cout_g + 127) / 128;
  const long long tp = (long long)a.batch * a.oh * a.ow / tpix;
  if (tp * a.tiles_co >= (1LL << 31)) return gg::fail(-2, "conv2d: too many tiles");
  a.tiles_pix = (int)tp;
  a.nslabs = a.cin_g / BKS;
  const long long blocks = tp * a.tiles_co * a.groups;
  int splitk = 1;
  if (blocks < split_at_patch()) {
    splitk = (int)((split_at_patch() + blocks - 1) / blocks);
    if (splitk > a.nslabs) splitk = a.nslabs;
    if (splitk < 1) splitk = 1;
  }
  a.slabs_per_split = (a.nslabs + splitk - 1) / splitk;
  a.splitk = (a.nslabs + a.slabs_per_split - 1) / a.slabs_per_split;
  if (a.splitk > 1) {       // every output element is covered by exactly one tile per split: no clearing needed
    if (int rc = splitk_prepare(a, a.splitk, false, st)) return rc;
  }
  dim3 grid((unsigned)(a.tiles_pix * a.tiles_co), (unsigned)a.splitk, (unsigned)a.groups);
  const bool sc = a.in_scale != nullptr;
  // the sign plane is written by the tile's own activation epilogue: not by split-K launches (their reduce pass applies
  // the activation)
  if (a.splitk > 1 || !a.act) a.sign_bits = nullptr;
  g_sign_bits_written = a.sign_bits ? 1 : 0;
  if (a.splitk > 1 || !a.act) a.amax_out = nullptr;
  g_amax_written = a.amax_out ? 1 : 0;
  NOTE_KERNEL("conv3x3_patch<limbs%d,%dpx,%dco,%s,%s>", limbs, tpix, narrow ? 64 : 128,
              a.mask_bits ? "bitmasked" : a.mask_ref ? "masked" : "plain", a.f16 ? "f16" : "bf16");
  if (a.mask_bits) {             // masked data gradient, the mask from the 1-bit sign plane (binary16 limbs)
    if (narrow && tpix == 256) {
      if (sc) conv3x3_patch_kernel<2, true, 256, 1, 2, 1, true><<<grid, 512, 0, st>>>(a, tw_log2);
      else conv3x3_patch_kernel<2, false, 256, 1, 2, 1, true><<<grid, 512, 0, st>>>(a, tw_log2);
    } else if (narrow) {
      if (sc) conv3x3_patch_kernel<2, true, 128, 1, 2, 1, true><<<grid, 256, 0, st>>>(a, tw_log2);
      else conv3x3_patch_kernel<2, false, 128, 1, 2, 1, true><<<grid, 256, 0, st>>>(a, tw_log2);
    } else if (tpix == 256) {
      if (sc) conv3x3_patch_kernel<2, true, 256, 2, 2, 1, true><<<grid, 512, 0, st>>>(a, tw_log2);
      else conv3x3_patch_kernel<2, false, 256, 2, 2, 1, true><<<grid, 512, 0, st>>>(a, tw_log2);
    } else {
      if (sc) conv3x3_patch_kernel<2, true, 128, 2, 2, 1, true><<<grid, 256, 0, st>>>(a, tw_log2);
      else conv3x3_patch_kernel<2, false, 128, 2, 2, 1, true><<<grid, 256, 0, st>>>(a, tw_log2);
    }
  } else if (a.mask_ref && a.f16) {     // masked data gradient on binary16 limbs (block exponents: any gradient magnitude)
    if (narrow && tpix == 256) {
      if (sc) conv3x3_patch_kernel<2, true, 256, 1, true, 1, true><<<grid, 512, 0, st>>>(a, tw_log2);
      else conv3x3_patch_kernel<2, false, 256, 1, true, 1, true><<<grid, 512, 0, st>>>(a, tw_log2);
    } else if (narrow) {
      if (sc) conv3x3_patch_kernel<2, true, 128, 1, true, 1, true><<<grid, 256, 0, st>>>(a, tw_log2);
      else conv3x3_patch_kernel<2, false, 128, 1, true, 1, true><<<grid, 256, 0, st>>>(a, tw_log2);
    } else if (tpix == 256) {
      if (sc) conv3x3_patch_kernel<2, true, 256, 2, true, 1, true><<<grid, 512, 0, st>>>(a, tw_log2);
      else conv3x3_patch_kernel<2, false, 256, 2, true, 1, true><<<grid, 512, 0, st>>>(a, tw_log2);
    } else {
      if (sc) conv3x3_patch_kernel<2, true, 128, 2, true, 1, true><<<grid, 256, 0, st>>>(a, tw_log2);
      else conv3x3_patch_kernel<2, false, 128, 2, true, 1, true><<<grid, 256, 0, st>>>(a, tw_log2);
    }
  } else if (a.mask_ref) {   // limbs == 2 (checked by the caller)
    if (narrow && tpix == 256) {
      if (sc) LIMBS12(limbs, conv3x3_patch_kernel<L, true, 256, 1, true><<<grid, 512, 0, st>>>(a, tw_log2));
      else LIMBS12(limbs, conv3x3_patch_kernel<L, false, 256, 1, true><<<grid, 512, 0, st>>>(a, tw_log2));
    } else if (narrow) {
      if (sc) LIMBS12(limbs, conv3x3_patch_kernel<L, true, 128, 1, true><<<grid, 256, 0, st>>>(a, tw_log2));
      else LIMBS12(limbs, conv3x3_patch_kernel<L, false, 128, 1, true><<<grid, 256, 0, st>>>(a, tw_log2));
    } else if (tpix == 256) {
      if (sc) LIMBS12(limbs, conv3x3_patch_kernel<L, true, 256, 2, true><<<grid, 512, 0, st>>>(a, tw_log2));
      else LIMBS12(limbs, conv3x3_patch_kernel<L, false, 256, 2, true><<<grid, 512, 0, st>>>(a, tw_log2));
    } else {
      if (sc) LIMBS12(limbs, conv3x3_patch_kernel<L, true, 128, 2, true><<<grid, 256, 0, st>>>(a, tw_log2));
      else LIMBS12(limbs, conv3x3_patch_kernel<L, false, 128, 2, true><<<grid, 256, 0, st>>>(a, tw_log2));
    }
  } else if (a.f16) {        // two binary16 limbs: the same four tile shapes
    if (narrow && tpix == 256) {
      if (sc) conv3x3_patch_kernel<2, true, 256, 1, false, 1, true><<<grid, 512, 0, st>>>(a, tw_log2);
      else conv3x3_patch_kernel<2, false, 256, 1, false, 1, true><<<grid, 512, 0, st>>>(a, tw_log2);
    } else if (narrow) {
      if (sc) conv3x3_patch_kernel<2, true, 128, 1, false, 1, true><<<grid, 256, 0, st>>>(a, tw_log2);
      else conv3x3_patch_kernel<2, false, 128, 1, false, 1, true><<<grid, 256, 0, st>>>(a, tw_log2);
    } else if (tpix == 256) {
      if (sc) conv3x3_patch_kernel<2, true, 256, 2, false, 1, true><<<grid, 512, 0, st>>>(a, tw_log2);
      else conv3x3_patch_kernel<2, false, 256, 2, false, 1, true><<<grid, 512, 0, st>>>(a, tw_log2);
    } else {
      if (sc) conv3x3_patch_kernel<2, true, 128, 2, false, 1, true><<<grid, 256, 0, st>>>(a, tw_log2);
      else conv3x3_patch_kernel<2, false, 128, 2, false, 1, true><<<grid, 256, 0, st>>>(a, tw_log2);
    }
  } else if (narrow && tpix == 256) {
    if (sc) LIMBS12(limbs, conv3x3_patch_kernel<L, true, 256, 1><<<grid, 512, 0, st>>>(a, tw_log2));
    else LIMBS12(limbs, conv3x3_patch_kernel<L, false, 256, 1><<<grid, 512, 0, st>>>(a, tw_log2));
  } else if (narrow) {
    if (sc) LIMBS12(limbs, conv3x3_patch_kernel<L, true, 128, 1><<<grid, 256, 0, st>>>(a, tw_log2));
    else LIMBS12(limbs, conv3x3_patch_kernel<L, false, 128, 1><<<grid, 256, 0, st>>>(a, tw_log2));
  } else if (limbs <= 2 && tpix == 256) {
    if (sc) LIMBS12(limbs, conv3x3_patch_kernel<L, true, 256><<<grid, 512, 0, st>>>(a, tw_log2));
    else LIMBS12(limbs, conv3x3_patch_kernel<L, false, 256><<<grid, 512, 0, st>>>(a, tw_log2));
  } else if (limbs <= 2) {
    if (sc) LIMBS12(limbs, conv3x3_patch_kernel<L, true, 128><<<grid, 256, 0, st>>>(a, tw_log2));
    else LIMBS12(limbs, conv3x3_patch_kernel<L, false, 128><<<grid, 256, 0, st>>>(a, tw_log2));
  } else {
    if (sc) conv3x3_patch_kernel<3, true, 128><<<grid, 256, 0, st>>>(a, tw_log2);
    else conv3x3_patch_kernel<3, false, 128><<<grid, 256, 0, st>>>(a, tw_log2);
  }
  const int rc = gg::launch_status("conv3x3_patch");
  if (rc || a.splitk <= 1) return rc;
  return splitk_reduce(a, a.splitk, st);                    // + out_scale / bias / fused activation
}

// 3x3 / stride 2 / pad 0 correlation on the patch-reuse tile of conv_s2_patch.hip (output width a multiple of 32):
// the generator's up-conv data gradients and the STN's down-sampling convolutions.  GG_S2_PATCH: measurement switch
// (0 = the generic re-gathering kernel as before, 128 / 256 = output pixels per block, 1 = chosen per launch: default).
static int s2_patch_tpix() {
  static const int v = env_int("GG_S2_PATCH", 1);
  return v;
}
int launch_conv_s2_patch(ConvArgs a, int stride, int tpix, hipStream_t st) {
  a.mh = a.oh; a.mw = a.ow;
  a.tile_pixels = tpix;
  a.tiles_co = (a.cout_g + 127) / 128;
  const long long tp = (long long)a.batch * a.oh * a.ow / tpix;
  if (tp * a.tiles_co >= (1LL << 31)) return gg::fail(-2, "conv2d: too many tiles");
  a.tiles_pix = (int)tp;
  a.nslabs = a.cin_g / 16;                     // 16-channel chunks
  const long long blocks = tp * a.tiles_co * a.groups;
  int splitk = 1;
  if (blocks < split_at_patch()) {
    splitk = (int)((split_at_patch() + blocks - 1) / blocks);
    const int max_split = a.nslabs / 2 > 0 ? a.nslabs / 2 : 1;      // >= 32 channels x 9 taps per split
    if (splitk > max_split) splitk = max_split;
    if (splitk < 1) splitk = 1;
  }
  a.slabs_per_split = (a.nslabs + splitk - 1) / splitk;
  a.splitk = (a.nslabs + a.slabs_per_split - 1) / a.slabs_per_split;
  if (a.splitk > 1) {       // every output element is covered by exactly one tile per split: no clearing needed
    if (int rc = splitk_prepare(a, a.splitk, false, st)) return rc;
  }
  dim3 grid((unsigned)(a.tiles_pix * a.tiles_co), (unsigned)a.splitk, (unsigned)a.groups);
  NOTE_KERNEL("conv3x3s%d_patch16<limbs2,%dpx,%s>", stride, tpix, a.f16 ? "f16" : "bf16");
  s2_patch_launch(a, stride, tpix, grid, st);
  const int rc = gg::launch_status("conv3x3s2_patch");
  if (rc || a.splitk <= 1) return rc;
  return splitk_reduce(a, a.splitk, st);                    // + out_scale / bias / fused activation
}

// all-classes transposed 3x3 / stride 2 kernel: power-of-two input width >= 4
int launch_convT_patch(ConvArgs a, int limbs, int pad, hipStream_t st) {
  int tw_log2 = 0;
  const int tw = a.w < 64 ? a.w : 64;
  while ((1 << tw_log2) < tw) ++tw_log2;
  a.tiles_co = (a.cout_g + 127) / 128;
  auto tiles_for = [&](int tq, int& tiles_y, int& edge) {
    const int th = tq >> tw_log2;
    tiles_y = (a.h + th) / th;                  // ceil((h + 1) / th)
    edge = (a.h + tq) / tq;                     // ceil((h + 1) / tq)
    return (long long)a.batch * (tiles_y * (a.w >> tw_log2) + edge);
  };
  int tiles_y, edge;
  int tq = 128;
  long long tp = tiles_for(tq, tiles_y, edge);
  // Round 5: the 16-channel-chunk tile of conv_t_c16.hip (64 co x 128 q on four waves, two blocks per CU).
  // GG_CONVT16: measurement switch (0 = the 32-channel-chunk tiles below as in round 4, 1 = per launch (default),
  // 64 / 128 = that tile's 64 / 128 co form on every two-limb launch it can serve).
  const int t16_mode = tuning(kTunConvT16);
  const int t16_tw = tuning(kTunConvT16Tw);                           // measurement override: tile width (power of two)
  if (t16_mode != 0 && limbs == 2 && t16_serves(a)) {
    const int tco = t16_mode == 128 ? 128 : 64;
    const int tiles_co = (a.cout_g + tco - 1) / tco;
    int tw16_log2 = tw_log2;
    while ((1 << tw16_log2) > t16_tw && tw16_log2 > 2) --tw16_log2;
    const int tw_keep = tw_log2;
    tw_log2 = tw16_log2;
    int ty16, edge16;
    const long long tp16 = tiles_for(128, ty16, edge16);
    tw_log2 = tw_keep;
    const long long blocks16 = tp16 * tiles_co * a.groups;
    // per launch (measured, profiles/r05_a_convt16_ab.txt): the new tile everywhere except the 4^2 -> 9^2 layer, whose
    // 25-position q-grid fills a fifth of a 128-q tile (0.063 -> 0.089 ms); 64-q tiles of the round-4 kernel there
    if (t16_mode != 1 || a.w >= 8) {
      if (tp16 * tiles_co >= (1LL << 31)) return gg::fail(-2, "conv2d: too many tiles");
      a.tiles_co = tiles_co;
      a.tiles_pix = (int)tp16;
      a.nslabs = a.cin_g / 16;                   // 16-channel chunks
      int splitk = 1;
      if (blocks16 < split_at_convt()) {
        splitk = (int)((gg::kNumCu + blocks16 - 1) / blocks16);
        const int max_split = a.nslabs / 2 > 0 ? a.nslabs / 2 : 1;      // >= 32 channels x 9 taps per split
        if (splitk > max_split) splitk = max_split;
        if (splitk < 1) splitk = 1;
      }
      a.slabs_per_split = (a.nslabs + splitk - 1) / splitk;
      a.splitk = (a.nslabs + a.slabs_per_split - 1) / a.slabs_per_split;
      const bool covered16 = a.oh - 1 <= 2 * a.h + 1 - pad && a.ow - 1 <= 2 * a.w + 1 - pad;
      if (a.splitk > 1) {
        if (int rc = splitk_prepare(a, a.splitk, !covered16, st)) return rc;
      } else if (!covered16) {
        const size_t out_elems = (size_t)a.batch * a.groups * a.cout_g * a.oh * a.ow;
        hipError_t e = hipMemsetAsync(a.y, 0, sizeof(float) * out_elems, st);
        if (e != hipSuccess) return gg::fail((int)e, "conv2d: memset failed");
      }
      dim3 grid16((unsigned)(a.tiles_pix * a.tiles_co), (unsigned)a.splitk, (unsigned)a.groups);
      NOTE_KERNEL("convT3x3s2_c16<limbs2,%dco,128q,%s%s>", tco, a.f16 ? "f16" : "bf16", a.xlimb ? ",prelimb" : "");
      if (a.xlimb) g_prelimb_served = 1;
      t16_launch(a, tco, tw16_log2, ty16, edge16, pad, grid16, st);
      const int rc = gg::launch_status("convT3x3s2_c16");
      if (rc || a.splitk <= 1) return rc;
      return splitk_reduce(a, a.splitk, st);
    }
  }
  static const bool force64 = getenv("GG_CONVT_TQ64") != nullptr;     // measurement switch
  if (force64 || limbs > 2 || tp * a.tiles_co * a.groups < 2 * gg::kNumCu) {
    tq = 64;
    tp = tiles_for(tq, tiles_y, edge);
  }
  if (tp * a.tiles_co >= (1LL << 31)) return gg::fail(-2, "conv2d: too many tiles");
  a.tiles_pix = (int)tp;
  a.nslabs = a.cin_g / BKS;
  const long long blocks = tp * a.tiles_co * a.groups;
  int splitk = 1;
  if (blocks < split_at_convt()) {
    splitk = (int)((gg::kNumCu + blocks - 1) / blocks);
    if (splitk > a.nslabs) splitk = a.nslabs;
    if (splitk < 1) splitk = 1;
  }
  a.slabs_per_split = (a.nslabs + splitk - 1) / splitk;
  a.splitk = (a.nslabs + a.slabs_per_split - 1) / a.slabs_per_split;
  // the q-grid's classes cover output rows / columns [-pad, 2 * size + 1 - pad]; anything beyond (output_padding
  // with pad > 0) receives no contribution and has to read as zero
  const bool covered = a.oh - 1 <= 2 * a.h + 1 - pad && a.ow - 1 <= 2 * a.w + 1 - pad;
  if (a.splitk > 1) {
    if (int rc = splitk_prepare(a, a.splitk, !covered, st)) return rc;
  } else if (!covered) {
    const size_t out_elems = (size_t)a.batch * a.groups * a.cout_g * a.oh * a.ow;
    hipError_t e = hipMemsetAsync(a.y, 0, sizeof(float) * out_elems, st);
    if (e != hipSuccess) return gg::fail((int)e, "conv2d: memset failed");
  }
  dim3 grid((unsigned)(a.tiles_pix * a.tiles_co), (unsigned)a.splitk, (unsigned)a.groups);
  const bool sc = a.in_scale != nullptr;
  NOTE_KERNEL("convT3x3s2_patch<limbs%d,%dq,%s>", limbs, tq, a.f16 ? "f16" : "bf16");
  if (a.f16) {
    if (tq == 128) {
      if (sc) convT3x3s2_patch_kernel<2, true, 128, true><<<grid, 512, 0, st>>>(a, tw_log2, tiles_y, edge, pad);
      else convT3x3s2_patch_kernel<2, false, 128, true><<<grid, 512, 0, st>>>(a, tw_log2, tiles_y, edge, pad);
    } else {
      if (sc) convT3x3s2_patch_kernel<2, true, 64, true><<<grid, 256, 0, st>>>(a, tw_log2, tiles_y, edge, pad);
      else convT3x3s2_patch_kernel<2, false, 64, true><<<grid, 256, 0, st>>>(a, tw_log2, tiles_y, edge, pad);
    }
  } else if (limbs <= 2 && tq == 128) {
    if (sc) LIMBS12(limbs, convT3x3s2_patch_kernel<L, true, 128><<<grid, 512, 0, st>>>(a, tw_log2, tiles_y, edge, pad));
    else LIMBS12(limbs, convT3x3s2_patch_kernel<L, false, 128><<<grid, 512, 0, st>>>(a, tw_log2, tiles_y, edge, pad));
  } else if (limbs <= 2) {
    if (sc) LIMBS12(limbs, convT3x3s2_patch_kernel<L, true, 64><<<grid, 256, 0, st>>>(a, tw_log2, tiles_y, edge, pad));
    else LIMBS12(limbs, convT3x3s2_patch_kernel<L, false, 64><<<grid, 256, 0, st>>>(a, tw_log2, tiles_y, edge, pad));
  } else {
    if (sc) convT3x3s2_patch_kernel<3, true, 64><<<grid, 256, 0, st>>>(a, tw_log2, tiles_y, edge, pad);
    else convT3x3s2_patch_kernel<3, false, 64><<<grid, 256, 0, st>>>(a, tw_log2, tiles_y, edge, pad);
  }
  const int rc = gg::launch_status("convT3x3s2_patch");
  if (rc || a.splitk <= 1) return rc;
  return splitk_reduce(a, a.splitk, st);
}

// ------------------------------------------------------------------------------------------------
// 1x1 convolution with <= 4 output channels (ToRGB, networks.py:353-372: 128..512 -> 3): a streaming reduction over
// the input channels - every input element is read exactly once, 16 B per lane, and meets NCO FMAs.  HBM-bound
// (4 * Cin B per pixel in, 4 * NCO out); on the 32-wide MFMA tile the same layer reached 3.0 TB/s.
// Block = 256 pixels of one image (64 lanes x 4 consecutive pixels) x 4 waves splitting the channels; the waves'
// partial sums meet in LDS.  wmat: [ci][co] (fp32 GEMM layout of pack_weight_kernel), in_scale: (N, Cin) or null.
// ------------------------------------------------------------------------------------------------
constexpr int FEWOUT_MAX_CIN = 512;
template <int NCO>
__global__ __launch_bounds__(256) void conv1x1_fewout_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                             const float* __restrict__ wmat,
                                                             const float* __restrict__ in_scale,
                                                             const float* __restrict__ out_scale,
                                                             const float* __restrict__ bias, int cin, long long hw) {
  __shared__ float sw[NCO][FEWOUT_MAX_CIN];
  __shared__ float4 red[4][NCO][64];
  const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, g = tid >> 6;
  for (int i = tid; i < cin * NCO; i += 256) {
    const int ci = i / NCO, j = i - ci * NCO;
    sw[j][ci] = wmat[i] * (in_scale ? in_scale[(size_t)n * cin + ci] : 1.f);
  }
  __syncthreads();
  const long long p = (long long)blockIdx.x * 256 + lane * 4;
  const bool ok = p < hw;                                   // hw % 4 == 0: a lane's four pixels are in or out together
  float4 acc[NCO];
#pragma unroll
  for (int j = 0; j < NCO; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) {
    const int per = (cin + 3) / 4, k0 = g * per, k1 = (k0 + per < cin) ? k0 + per : cin;
    const float* src = x + ((size_t)n * cin + k0) * hw + p;
#pragma unroll 8
    for (int k = k0; k < k1; ++k, src += hw) {
      const float4 v = *reinterpret_cast<const float4*>(src);
#pragma unroll
      for (int j = 0; j < NCO; ++j) {
        const float wj = sw[j][k];
        acc[j].x += v.x * wj; acc[j].y += v.y * wj; acc[j].z += v.z * wj; acc[j].w += v.w * wj;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NCO; ++j) red[g][j][lane] = acc[j];
  __syncthreads();
  if (g < NCO && ok) {                                      // wave j finishes output channel j
    const int j = g;
    float4 r = red[0][j][lane];
#pragma unroll
    for (int q = 1; q < 4; ++q) {
      const float4 t = red[q][j][lane];
      r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
    }
    const float sc = out_scale ? out_scale[(size_t)n * NCO + j] : 1.f, bi = bias ? bias[j] : 0.f;
    r.x = r.x * sc + bi; r.y = r.y * sc + bi; r.z = r.z * sc + bi; r.w = r.w * sc + bi;
    *reinterpret_cast<float4*>(y + ((size_t)n * NCO + j) * hw + p) = r;
  }
}

// 1x1 convolution with <= 4 INPUT channels (the data gradient of the last ToRGB layer: 3 -> 128 channels at the output
// resolution): every output element is a 3-term dot product - a write-only stream (537 MB at 256^2, batch 16) that the
// 32-wide MFMA tile served at 2.7 TB/s with K = 3 of its 16 reduction slots in use.  One block = 1024 pixels x 16 output
// channels of one image; x is re-read from L2 by the blocks of the other channel groups (12 MB in all).
constexpr int FEWIN_CO = 16;
template <int NCI>
__global__ __launch_bounds__(256) void conv1x1_fewin_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                            const float* __restrict__ wmat,
                                                            const float* __restrict__ in_scale,
                                                            const float* __restrict__ out_scale,
                                                            const float* __restrict__ bias, int cout, long long hw,
                                                            int nt) {
  __shared__ float sw[FEWIN_CO][NCI + 1];                  // [co][ci] style-scaled weights, [co][NCI] = bias
  __shared__ float ssc[FEWIN_CO];
  const int n = blockIdx.z, co0 = blockIdx.y * FEWIN_CO, tid = threadIdx.x;
  if (tid < FEWIN_CO * NCI) {
    const int j = tid / NCI, ci = tid - j * NCI, co = co0 + j;
    sw[j][ci] = co < cout ? wmat[ci * cout + co] * (in_scale ? in_scale[(size_t)n * NCI + ci] : 1.f) : 0.f;
  }
  if (tid < FEWIN_CO) {
    const int co = co0 + tid;
    sw[tid][NCI] = (bias && co < cout) ? bias[co] : 0.f;
    ssc[tid] = (out_scale && co < cout) ? out_scale[(size_t)n * cout + co] : 1.f;
  }
  __syncthreads();
  const long long p = ((long long)blockIdx.x * 256 + tid) * 4;
  if (p >= hw) return;                                      // hw % 4 == 0
  float4 v[NCI];
#pragma unroll
  for (int ci = 0; ci < NCI; ++ci) v[ci] = *reinterpret_cast<const float4*>(x + ((size_t)n * NCI + ci) * hw + p);
  const int jn = cout - co0 < FEWIN_CO ? cout - co0 : FEWIN_CO;
  for (int j = 0; j < jn; ++j) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ci = 0; ci < NCI; ++ci) {
      const float wj = sw[j][ci];
      r.x += v[ci].x * wj; r.y += v[ci].y * wj; r.z += v[ci].z * wj; r.w += v[ci].w * wj;
    }
    const float sc = ssc[j], bi = sw[j][NCI];
    r.x = r.x * sc + bi; r.y = r.y * sc + bi; r.z = r.z * sc + bi; r.w = r.w * sc + bi;
    float* dst = y + ((size_t)n * cout + co0 + j) * hw + p;
    if (nt) __builtin_nontemporal_store(f32x4{r.x, r.y, r.z, r.w}, reinterpret_cast<f32x4*>(dst));
    else *reinterpret_cast<float4*>(dst) = r;
  }
}

bool fewin_serves(const ConvArgs& a, int stride, int pad, int mode) {
  const long long hw = (long long)a.h * a.w;
  static const bool off = getenv("GG_NO_FEWIN") != nullptr;       // measurement switch
  return !off && mode == 0 && stride == 1 && pad == 0 && a.groups == 1 && a.cin_g >= 1 && a.cin_g <= 4 && a.cout_g >= 16 &&
         a.wmat && hw % 4 == 0 && hw >= 4096 && !a.act && !a.mask_ref && a.batch <= 65535 &&
         (a.cout_g + FEWIN_CO - 1) / FEWIN_CO <= 65535 &&
         (reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.y) & 15) == 0;
}

int launch_conv1x1_fewin(const ConvArgs& a, hipStream_t st) {
  NOTE_KERNEL("conv1x1_fewin");
  const long long hw = (long long)a.h * a.w;
  dim3 grid((unsigned)((hw / 4 + 255) / 256), (unsigned)((a.cout_g + FEWIN_CO - 1) / FEWIN_CO), (unsigned)a.batch);
  switch (a.cin_g) {
    case 1: conv1x1_fewin_kernel<1><<<grid, 256, 0, st>>>(a.y, a.x, a.wmat, a.in_scale, a.out_scale, a.bias, a.cout_g, hw, a.nt_store); break;
    case 2: conv1x1_fewin_kernel<2><<<grid, 256, 0, st>>>(a.y, a.x, a.wmat, a.in_scale, a.out_scale, a.bias, a.cout_g, hw, a.nt_store); break;
    case 3: conv1x1_fewin_kernel<3><<<grid, 256, 0, st>>>(a.y, a.x, a.wmat, a.in_scale, a.out_scale, a.bias, a.cout_g, hw, a.nt_store); break;
    default: conv1x1_fewin_kernel<4><<<grid, 256, 0, st>>>(a.y, a.x, a.wmat, a.in_scale, a.out_scale, a.bias, a.cout_g, hw, a.nt_store); break;
  }
  return gg::launch_status("conv1x1_fewin");
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution with <= 4 INPUT channels and its bias + (leaky) ReLU: the perceptual trunk's RGB stem
// (3 -> 64 @128^2 over 32 images; lpips_backbones.py:109).  K = 27: on the fp32 MFMA tile the launch was a 134 MB
// scalar-store stream at 1.8 TB/s (74 us) followed by a separate activation pass (42 us).  Here a lane owns 4 consecutive
// pixels of one image row, keeps the 3 x 6 x NCI input window in registers and walks the block's 32 output channels with
// the 9 * NCI weights broadcast from LDS (108 FMAs per channel): one float4 store per channel, the activation in registers,
// and - for the layer's masked data gradient - bit j of the lane's four words = (stored output of channel co0 + j > 0),
// the sign plane of conv_common.h (ConvArgs::sign_bits).  Exact fp32 products, k ascending.
// wmat: [(ci, ky, kx)][co] (fp32 GEMM layout of pack_weight_kernel).
// ------------------------------------------------------------------------------------------------
constexpr int FEWIN3_CO = 32;
template <int NCI>
__global__ __launch_bounds__(256) void conv3x3_fewin_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                            const float* __restrict__ wmat,
                                                            const float* __restrict__ bias, int cout, int h, int w,
                                                            int act, float alpha, float gain,
                                                            unsigned* __restrict__ sign_bits, int bit_words, int nt) {
  __shared__ float sw[FEWIN3_CO][NCI * 9 + 1];             // [co][k], [co][NCI * 9] = bias
  const int n = blockIdx.z, co0 = blockIdx.y * FEWIN3_CO, tid = threadIdx.x;
  const long long hw = (long long)h * w;
  for (int i = tid; i < FEWIN3_CO * (NCI * 9 + 1); i += 256) {
    const int j = i / (NCI * 9 + 1), k = i - j * (NCI * 9 + 1), co = co0 + j;
    sw[j][k] = co < cout ? (k < NCI * 9 ? wmat[(size_t)k * cout + co] : (bias ? bias[co] : 0.f)) : 0.f;
  }
  __syncthreads();
  const long long p = ((long long)blockIdx.x * 256 + tid) * 4;
  if (p >= hw) return;                                      // w % 4 == 0: the four pixels share a row
  const int py = (int)(p / w), px = (int)(p - (long long)py * w);
  const bool left = px > 0, right = px + 4 < w;
  float v[NCI][3][6];
#pragma unroll
  for (int ci = 0; ci < NCI; ++ci) {
    const float* plane = x + ((size_t)n * NCI + ci) * hw;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int yy = py + dy - 1;
      if ((unsigned)yy < (unsigned)h) {
        const float* row = plane + (size_t)yy * w + px;
        const float4 c = *reinterpret_cast<const float4*>(row);
        v[ci][dy][0] = left ? row[-1] : 0.f;
        v[ci][dy][1] = c.x; v[ci][dy][2] = c.y; v[ci][dy][3] = c.z; v[ci][dy][4] = c.w;
        v[ci][dy][5] = right ? row[4] : 0.f;
      } else {
#pragma unroll
        for (int q = 0; q < 6; ++q) v[ci][dy][q] = 0.f;
      }
    }
  }
  unsigned sgn[4] = {0u, 0u, 0u, 0u};
  const int jn = cout - co0 < FEWIN3_CO ? cout - co0 : FEWIN3_CO;
  for (int j = 0; j < jn; ++j) {
    float r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const float wj = sw[j][(ci * 3 + dy) * 3 + dx];
#pragma unroll
          for (int q = 0; q < 4; ++q) r[q] = fmaf(v[ci][dy][dx + q], wj, r[q]);
        }
    const float bi = sw[j][NCI * 9];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float t = r[q] + bi;
      if (act) t = (t > 0.f ? t : t * alpha) * gain;
      r[q] = t;
      sgn[q] |= (t > 0.f ? 1u : 0u) << j;
    }
    float* dst = y + ((size_t)n * cout + co0 + j) * hw + p;
    if (nt) __builtin_nontemporal_store(f32x4{r[0], r[1], r[2], r[3]}, reinterpret_cast<f32x4*>(dst));
    else *reinterpret_cast<float4*>(dst) = make_float4(r[0], r[1], r[2], r[3]);
  }
  if (sign_bits) {
    unsigned* dst = sign_bits + ((size_t)n * hw + p) * bit_words + blockIdx.y;
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[(size_t)q * bit_words] = sgn[q];
  }
}

// smallest batch * H * W the few-input-channel 3x3 kernel takes (GG_FEWIN3_MIN: measurement override; at 2048 it would also
// take the data gradient of the flow head's last layer, 2 -> 512 @16^2: 13 us against the fp32 MFMA tile's 14, session 33)
static long long fewin3_min_pixels() {
  static const long long v = env_int("GG_FEWIN3_MIN", 65536);
  return v;
}
bool fewin3_serves(const ConvArgs& a, int stride, int pad, int mode) {
  static const bool off = getenv("GG_NO_FEWIN3") != nullptr;      // measurement switch
  const long long hw = (long long)a.h * a.w;
  return !off && mode == 0 && stride == 1 && pad == 1 && a.groups == 1 && a.cin_g >= 1 && a.cin_g <= 4 && a.cout_g >= 16 &&
         a.wmat && !a.in_scale && !a.out_scale && !a.mask_ref && !a.mask_bits && !(a.act && a.act_noise) &&
         a.w % 4 == 0 && (long long)a.batch * hw >= fewin3_min_pixels() && a.batch <= 65535 &&
         (a.cout_g + FEWIN3_CO - 1) / FEWIN3_CO <= 65535 && (!a.sign_bits || a.cout_g % 32 == 0) &&
         (reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.y) & 15) == 0;
}

int launch_conv3x3_fewin(const ConvArgs& a, hipStream_t st) {
  NOTE_KERNEL("conv3x3_fewin%s", a.act ? "+bias+lrelu" : "");
  const long long hw = (long long)a.h * a.w;
  dim3 grid((unsigned)((hw / 4 + 255) / 256), (unsigned)((a.cout_g + FEWIN3_CO - 1) / FEWIN3_CO), (unsigned)a.batch);
  const float* bias = a.act ? a.act_bias : a.bias;
#define GG_FEWIN3(N) conv3x3_fewin_kernel<N><<<grid, 256, 0, st>>>(a.y, a.x, a.wmat, bias, a.cout_g, a.h, a.w, a.act, \
                                                                   a.act_alpha, a.act_gain, a.sign_bits, a.bit_words, a.nt_store)
  switch (a.cin_g) {
    case 1: GG_FEWIN3(1); break;
    case 2: GG_FEWIN3(2); break;
    case 3: GG_FEWIN3(3); break;
    default: GG_FEWIN3(4); break;
  }
#undef GG_FEWIN3
  g_sign_bits_written = a.sign_bits ? 1 : 0;
  return gg::launch_status("conv3x3_fewin");
}

bool fewout_serves(const ConvArgs& a, int stride, int pad, int mode) {
  const long long hw = (long long)a.h * a.w;
  static const bool off = getenv("GG_NO_FEWOUT") != nullptr;      // measurement switch
  return !off && mode == 0 && stride == 1 && pad == 0 && a.groups == 1 && a.cout_g >= 1 && a.cout_g <= 4 && a.wmat &&
         a.cin_g <= FEWOUT_MAX_CIN && hw % 4 == 0 && !a.act && !a.mask_ref && a.batch <= 65535 &&
         (reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.y) & 15) == 0;
}

int launch_conv1x1_fewout(const ConvArgs& a, hipStream_t st) {
  NOTE_KERNEL("conv1x1_fewout");
  const long long hw = (long long)a.h * a.w;
  dim3 grid((unsigned)((hw + 255) / 256), (unsigned)a.batch);
  switch (a.cout_g) {
    case 1: conv1x1_fewout_kernel<1><<<grid, 256, 0, st>>>(a.y, a.x, a.wmat, a.in_scale, a.out_scale, a.bias, a.cin_g, hw); break;
    case 2: conv1x1_fewout_kernel<2><<<grid, 256, 0, st>>>(a.y, a.x, a.wmat, a.in_scale, a.out_scale, a.bias, a.cin_g, hw); break;
    case 3: conv1x1_fewout_kernel<3><<<grid, 256, 0, st>>>(a.y, a.x, a.wmat, a.in_scale, a.out_scale, a.bias, a.cin_g, hw); break;
    default: conv1x1_fewout_kernel<4><<<grid, 256, 0, st>>>(a.y, a.x, a.wmat, a.in_scale, a.out_scale, a.bias, a.cin_g, hw); break;
  }
  return gg::launch_status("conv1x1_fewout");
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution with <= 4 output channels: the data gradient of the perceptual trunk's RGB stem
// (64 -> 3 @128^2 over 32 images: 0.40 ms on the 32-wide MFMA tile, whose other 29 output columns are padding) and the
// flow head's last layer (512 -> 2).  A streaming VALU reduction like conv1x1_fewout_kernel: a lane owns 4 consecutive
// pixels of one image row (W % 4 == 0), reads per input channel the three rows around them (one aligned float4 plus
// the two neighbouring pixels each; rows / columns outside the image read as zero) and applies the 9 taps; the four
// waves of a block split the input channels and meet in LDS.  HBM-bound: 4 * Cin bytes per pixel in.
// wmat: [(ci, ky, kx)][co] (fp32 GEMM layout of pack_weight_kernel).
// ------------------------------------------------------------------------------------------------
// MASKB (round 6): x is the gradient of a conv + leaky-ReLU layer's OUTPUT and the activation's backward is applied as it is
// read - x * (bit ? gain : alpha * gain), bit (k & 31) of word [(n * H * W + pixel) * bit_words + k / 32] of the sign plane
// the layer's forward wrote (ConvArgs::sign_bits): 18 words per lane and 32 channels instead of a separate masked-gradient
// pass over the 134 MB tensor (the stem's ReLU backward: 57 us).
template <int NCO, bool MASKB = false>
__global__ __launch_bounds__(256) void conv3x3_fewout_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                             const float* __restrict__ wmat,
                                                             const float* __restrict__ in_scale,
                                                             const float* __restrict__ out_scale,
                                                             const float* __restrict__ bias, int cin, int h, int w,
                                                             const unsigned* __restrict__ mask_bits = nullptr,
                                                             int bit_words = 0, float mask_alpha = 0.f,
                                                             float mask_gain = 1.f) {
  extern __shared__ __attribute__((aligned(16))) float sw[];                     // cin * 9 * NCO style-scaled weights
  __shared__ float4 red[4][NCO][64];
  const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, g = tid >> 6;
  const long long hw = (long long)h * w;
  for (int i = tid; i < cin * 9 * NCO; i += 256) {
    const int ci = i / (9 * NCO);
    sw[i] = wmat[i] * (in_scale ? in_scale[(size_t)n * cin + ci] : 1.f);
  }
  __syncthreads();
  const long long p = (long long)blockIdx.x * 256 + lane * 4;
  const bool ok = p < hw;
  float4 acc[NCO];
#pragma unroll
  for (int j = 0; j < NCO; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) {
    const int py = (int)(p / w), px = (int)(p - (long long)py * w);
    const int per = (cin + 3) / 4, k0 = g * per, k1 = (k0 + per < cin) ? k0 + per : cin;
    const bool left = px > 0, right = px + 4 < w;
    unsigned wd[3][6];                                       // MASKB: sign words of the 3 x 6 window, current 32-channel group
    const float m_neg = mask_alpha * mask_gain;
    for (int k = k0; k < k1; ++k) {
      const float* plane = x + ((size_t)n * cin + k) * hw;
      const float* wk = sw + (size_t)k * 9 * NCO;
      if (MASKB && (k == k0 || (k & 31) == 0)) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int yy = py + dy - 1;
          const bool rok = (unsigned)yy < (unsigned)h;
          const unsigned* brow = mask_bits + ((size_t)n * hw + (size_t)(rok ? yy : py) * w + px) * bit_words + (k >> 5);
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            const bool cok = rok && (q == 0 ? left : (q == 5 ? right : true));
            wd[dy][q] = cok ? brow[(long long)(q - 1) * bit_words] : 0u;
          }
        }
      }
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int yy = py + dy - 1;
        if ((unsigned)yy >= (unsigned)h) continue;
        const float* row = plane + (size_t)yy * w + px;
        const float4 c = *reinterpret_cast<const float4*>(row);
        const float l = left ? row[-1] : 0.f, r = right ? row[4] : 0.f;
        float v[6] = {l, c.x, c.y, c.z, c.w, r};
        if (MASKB) {
#pragma unroll
          for (int q = 0; q < 6; ++q) v[q] *= ((wd[dy][q] >> (k & 31)) & 1u) ? mask_gain : m_neg;
        }
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
          for (int j = 0; j < NCO; ++j) {
            const float wj = wk[(dy * 3 + dx) * NCO + j];
            acc[j].x += v[dx] * wj; acc[j].y += v[dx + 1] * wj; acc[j].z += v[dx + 2] * wj; acc[j].w += v[dx + 3] * wj;
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NCO; ++j) red[g][j][lane] = acc[j];
  __syncthreads();
  if (g < NCO && ok) {                                      // wave j finishes output channel j
    const int j = g;
    float4 r = red[0][j][lane];
#pragma unroll
    for (int q = 1; q < 4; ++q) {
      const float4 t = red[q][j][lane];
      r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
    }
    const float sc = out_scale ? out_scale[(size_t)n * NCO + j] : 1.f, bi = bias ? bias[j] : 0.f;
    r.x = r.x * sc + bi; r.y = r.y * sc + bi; r.z = r.z * sc + bi; r.w = r.w * sc + bi;
    *reinterpret_cast<float4*>(y + ((size_t)n * NCO + j) * hw + p) = r;
  }
}

bool fewout3_serves(const ConvArgs& a, int stride, int pad, int mode) {
  static const bool off = getenv("GG_NO_FEWOUT") != nullptr;      // measurement switch
  // (small planes leave the chip empty: 512 -> 2 @16^2 at batch 16 is 16 blocks, 96 us here against 31 us on the MFMA tile)
  return !off && mode == 0 && stride == 1 && pad == 1 && a.groups == 1 && a.cout_g >= 1 && a.cout_g <= 4 && a.wmat &&
         (long long)a.batch * a.h * a.w >= 256LL * 512 && a.cin_g * 9 * a.cout_g <= 16384 && a.cin_g <= FEWOUT_MAX_CIN && a.w % 4 == 0 && !a.act && !a.mask_ref &&
         a.batch <= 65535 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.y) & 15) == 0;
}

int launch_conv3x3_fewout(const ConvArgs& a, hipStream_t st) {
  NOTE_KERNEL("conv3x3_fewout");
  const long long hw = (long long)a.h * a.w;
  dim3 grid((unsigned)((hw + 255) / 256), (unsigned)a.batch);
  const size_t smem = sizeof(float) * (size_t)a.cin_g * 9 * a.cout_g;
  switch (a.cout_g) {
    case 1: conv3x3_fewout_kernel<1><<<grid, 256, smem, st>>>(a.y, a.x, a.wmat, a.in_scale, a.out_scale, a.bias, a.cin_g, a.h, a.w); break;
    case 2: conv3x3_fewout_kernel<2><<<grid, 256, smem, st>>>(a.y, a.x, a.wmat, a.in_scale, a.out_scale, a.bias, a.cin_g, a.h, a.w); break;
    case 3: conv3x3_fewout_kernel<3><<<grid, 256, smem, st>>>(a.y, a.x, a.wmat, a.in_scale, a.out_scale, a.bias, a.cin_g, a.h, a.w); break;
    default: conv3x3_fewout_kernel<4><<<grid, 256, smem, st>>>(a.y, a.x, a.wmat, a.in_scale, a.out_scale, a.bias, a.cin_g, a.h, a.w); break;
  }
  return gg::launch_status("conv3x3_fewout");
}

constexpr int kNotFused = GG_NOT_SERVED;      // masked-input request that no kernel serves: nothing was launched

// the data gradient of a few-input-channel conv + (leaky) ReLU layer with the activation's backward read from the sign plane
int launch_conv3x3_fewout_masked(const ConvArgs& a, hipStream_t st) {
  NOTE_KERNEL("conv3x3_fewout+lrelu-mask(sign plane)");
  const long long hw = (long long)a.h * a.w;
  dim3 grid((unsigned)((hw + 255) / 256), (unsigned)a.batch);
  const size_t smem = sizeof(float) * (size_t)a.cin_g * 9 * a.cout_g;
#define GG_FEWOUT3M(N) conv3x3_fewout_kernel<N, true><<<grid, 256, smem, st>>>(a.y, a.x, a.wmat, a.in_scale, a.out_scale, \
                         a.bias, a.cin_g, a.h, a.w, a.mask_bits, a.bit_words, a.mask_alpha, a.mask_gain)
  switch (a.cout_g) {
    case 1: GG_FEWOUT3M(1); break;
    case 2: GG_FEWOUT3M(2); break;
    case 3: GG_FEWOUT3M(3); break;
    default: GG_FEWOUT3M(4); break;
  }
#undef GG_FEWOUT3M
  return gg::launch_status("conv3x3_fewout_masked");
}

template <int KS>
int conv_dispatch(ConvArgs a, int stride, int pad, int mode, hipStream_t st, int limbs = 0) {
  if (a.xlimb && a.xlimb_e) {
    // the operand exists in limb form only: exactly one kernel reads that (conv_t_c16.hip, 64 output channels per block)
    if (!(KS == 3 && limbs == 2 && a.f16 && mode == 1 && stride == 2 && pad <= 1 && !a.act && !a.in_scale && a.groups == 1 &&
          a.w >= 8 && (a.w & (a.w - 1)) == 0 && tuning(kTunConvT16) == 1 && t16_serves(a)))
      return kNotFused;
    return launch_convT_patch(a, limbs, pad, st);
  }
  if (a.mask_ref || a.mask_bits) {
    int tw_log2;
    if (!((limbs == 1 || limbs == 2) && KS == 3 && mode == 0 && stride == 1 && pad == 1)) return kNotFused;
    if (a.mask_bits && !(a.f16 && limbs == 2 && a.groups == 1)) return kNotFused;
    const long long tiles256 = (long long)a.batch * a.oh * a.ow / 256 * ((a.cout_g + 127) / 128) * a.groups;
    if (tiles256 >= 2 * gg::kNumCu && a.cin_g > patch256_min_cin() && patch_geometry(a, 256, tw_log2))
      return launch_conv_patch(a, limbs, tw_log2, 256, st);
    if (patch_geometry(a, 128, tw_log2)) return launch_conv_patch(a, limbs, tw_log2, 128, st);
    return kNotFused;
  }
  if (KS == 1 && limbs == 0 && fewout_serves(a, stride, pad, mode)) return launch_conv1x1_fewout(a, st);
  if (KS == 1 && limbs == 0 && fewin_serves(a, stride, pad, mode)) return launch_conv1x1_fewin(a, st);
  if (KS == 3 && limbs == 0 && fewout3_serves(a, stride, pad, mode)) return launch_conv3x3_fewout(a, st);
  if (KS == 3 && limbs == 0 && fewin3_serves(a, stride, pad, mode)) return launch_conv3x3_fewin(a, st);
  if (!a.act && limbs && KS == 3 && mode == 1 && pad <= 1 && a.w >= 4 && (a.w & (a.w - 1)) == 0 &&
      (long long)a.cin_g * a.h * a.w * 4 < (1LL << 31) && (long long)a.cout_g * a.oh * a.ow * 4 < (1LL << 31))
    return launch_convT_patch(a, limbs, pad, st);
  if (limbs && KS == 3 && mode == 0 && stride == 1 && pad == 1) {
    int tw_log2;
    // 256-pixel tiles when they still fill the chip (>= 2 blocks per CU), else 128-pixel tiles
    const long long tiles256 = (long long)a.batch * a.oh * a.ow / 256 * ((a.cout_g + 127) / 128) * a.groups;
    if (limbs <= 2 && tiles256 >= 2 * gg::kNumCu && a.cin_g > patch256_min_cin() && patch_geometry(a, 256, tw_log2))
      return launch_conv_patch(a, limbs, tw_log2, 256, st);
    // GG_C16_S1 (measurement switch): the 16-channel-chunk tile of conv_s2_patch.hip in its stride-1 form instead of the
    // 128-pixel variant of conv3x3_patch_kernel, for layers with more than 64 output channels
    static const int c16_s1 = env_int("GG_C16_S1", 0);
    if (c16_s1 && limbs == 2 && a.cout_g > 64 && s2_patch_serves(a, 128) &&
        (!a.act || (!a.act_noise || (reinterpret_cast<uintptr_t>(a.act_noise) & 15) == 0)))
      return launch_conv_s2_patch(a, 1, 128, st);
    if (patch_geometry(a, 128, tw_log2, limbs)) return launch_conv_patch(a, limbs, tw_log2, 128, st);
  }
  if (limbs == 2 && KS == 3 && mode == 0 && stride == 2 && pad == 0 && s2_patch_tpix() > 0 &&
      (!a.act || ((a.oh * a.ow) % 4 == 0 && (!a.act_noise || (reinterpret_cast<uintptr_t>(a.act_noise) & 15) == 0)))) {
    // 256-pixel (8-wave, one block per CU) tiles halve the weight stream per output: they win once the reduction is deep
    // and every CU still gets a tile (measured, batch 16: 512 -> 512 @65^2 -> 32^2 0.277 vs 0.293 ms, 256 -> 512 @129^2
    // 0.535 vs 0.588, 128 -> 256 @257^2 0.576 vs 0.621; but 64 -> 128 @129^2 0.067 vs 0.045 and 128 -> 512 @65^2 0.092
    // vs 0.075: profiles/r04_h_s2_patch_ab.txt).  GG_S2_PATCH = 128 / 256 forces one of them, 1 = this rule.
    const long long tiles256 = (long long)a.batch * a.oh * a.ow / 256 * ((a.cout_g + 127) / 128) * a.groups;
    int tpix = s2_patch_tpix() == 256 ? 256 : 128;
    if (s2_patch_tpix() == 1)
      tpix = ((tiles256 >= gg::kNumCu && a.cin_g >= 256) || (tiles256 >= 2 * gg::kNumCu && a.cin_g >= 128)) ? 256 : 128;
    if (tpix == 256 && !s2_patch_serves(a, 256)) tpix = 128;
    if (s2_patch_serves(a, tpix)) return launch_conv_s2_patch(a, 2, tpix, st);
  }
  // no other kernel carries the activation in its epilogue: the split-K reduce pass applies it when the launch has one,
  // a separate in-place pass otherwise (act_later)
  const bool act_later = a.act != 0;
  // tile selector.  (Measured on the 128->128 @256^2 layer: an 8-wave 128x128 variant, a 128co x 256pix
  // variant with 8 accumulators per wave and BK = 32 are all within -20..+1 % of this 4-wave tile.)
  const int narrow = a.cout_g <= 32 ? 1 : (a.cout_g <= 64 ? 2 : 0);
  ConvArgs plans[4];
  int nplans = 0;
  bool needs_zero = false;
  if (mode == 0) {
    a.mh = a.oh; a.mw = a.ow;
    a.ys = 1; a.yo = 0; a.xs = 1; a.xo = 0;
    a.bs = stride; a.byo = -pad; a.bxo = -pad;
    a.py = a.px = 0; a.nty = a.ntx = KS;
    a.ktot = a.cin_g * KS * KS;
    if (plan_conv(a, limbs ? split_tile(a, limbs) : narrow, limbs ? BKS : BK)) plans[nplans++] = a;
  } else {
    // transposed, stride 2: one dense sub-problem per parity class of u = y + pad
    for (int py = 0; py < 2; ++py) {
      for (int px = 0; px < 2; ++px) {
        ConvArgs c = a;
        const int nty = (KS - py + 1) / 2, ntx = (KS - px + 1) / 2;     // taps ky = py, py+2, ... < KS
        auto axis = [&](int p, int osize, int& q0, int& o0, int& cnt) {
          q0 = 0;                                                       // y = 2*q + p - pad >= 0
          while (2 * q0 + p - pad < 0) ++q0;
          o0 = 2 * q0 + p - pad;
          cnt = (o0 < osize) ? (osize - 1 - o0) / 2 + 1 : 0;
        };
        int qy0, yo, mh, qx0, xo, mw;
        axis(py, a.oh, qy0, yo, mh);
        axis(px, a.ow, qx0, xo, mw);
        if (mh <= 0 || mw <= 0) continue;
        if (nty <= 0 || ntx <= 0) { needs_zero = true; continue; }     // class receives no tap: stays 0
        c.mh = mh; c.mw = mw;
        c.ys = 2; c.yo = yo; c.xs = 2; c.xo = xo;
        c.bs = 1; c.byo = qy0; c.bxo = qx0;
        c.py = py; c.px = px; c.nty = nty; c.ntx = ntx;
        c.ktot = a.cin_g * nty * ntx;
        if (plan_conv(c, limbs ? split_tile(c, limbs) : narrow, limbs ? BKS : BK)) plans[nplans++] = c;
      }
    }
  }
  int max_split = 1;
  for (int i = 0; i < nplans; ++i) max_split = plans[i].splitk > max_split ? plans[i].splitk : max_split;
  if (max_split > 1) {
    // split-K somewhere: EVERY plan writes raw partials (its splits of the scratch copies; parity-class plans write
    // disjoint output positions), then one reduce pass adds the copies in order and applies the epilogue.  With more
    // than one plan (or a class that receives no tap) a copy is not fully covered by its writers: cleared first.
    if (int rc = splitk_prepare(a, max_split, mode == 1, st)) return rc;
    for (int i = 0; i < nplans; ++i) { plans[i].part = a.part; plans[i].part_stride = a.part_stride; }
  } else if (needs_zero) {   // a parity class without taps stays zero
    const size_t out_elems = (size_t)a.batch * a.groups * a.cout_g * a.oh * a.ow;
    hipError_t e = hipMemsetAsync(a.y, 0, sizeof(float) * out_elems, st);
    if (e != hipSuccess) return gg::fail((int)e, "conv2d: memset failed");
  }
  for (int i = 0; i < nplans; ++i) {
    int rc;
    if (limbs) rc = (mode == 0) ? launch_conv_split<KS, 0>(plans[i], limbs, st) : launch_conv_split<KS, 1>(plans[i], limbs, st);
    else rc = (mode == 0) ? launch_conv<KS, 0>(plans[i], narrow, st) : launch_conv<KS, 1>(plans[i], narrow, st);
    if (rc) return rc;
  }
  if (max_split > 1) return splitk_reduce(a, max_split, st);       // + out_scale / bias / activation / residual
  return act_later ? post_activation(a, st) : 0;
}

}  // namespace

extern "C" const char* gg_last_conv_kernel(void) { return g_last_kernel; }
extern "C" int gg_last_sign_bits_written(void) { return g_sign_bits_written; }
extern "C" int gg_last_amax_written(void) { return g_amax_written; }

extern "C" int gg_set_tuning(const char* name, int value) {
  if (!name) return gg::fail(-2, "set_tuning: null name");
  for (int i = 0; i < kTunCount; ++i)
    if (strcmp(name, g_tunables[i].name) == 0) {
      g_tunables[i].value = value;
      g_tunables[i].init = value != GG_TUNING_RESET;      // reset: environment / built-in default at the next use
      return 0;
    }
  return gg::fail(-2, "set_tuning: unknown switch %s", name);
}

extern "C" int gg_conv_pack_weight_f32(float* wmat, const float* w, int groups, int cout_g, int cin_g, int kh,
                                       int kw, int transpose_io, int flip, float scale, void* stream) {
  const long long total = (long long)groups * cout_g * cin_g * kh * kw;
  if (total <= 0) return 0;
  if (!wmat || !w) return gg::fail(-2, "conv_pack_weight: null pointer");
  pack_weight_kernel<<<gg::stream_grid(total, 256), 256, 0, gg::as_stream(stream)>>>(wmat, w, total, cout_g, cin_g,
                                                                                    kh, kw, transpose_io, flip,
                                                                                    scale);
  return gg::launch_status("conv_pack_weight");
}

namespace {
struct MaskArgs {
  const float* ref = nullptr;
  float alpha = 0.f, gain = 1.f;
  const unsigned* bits = nullptr;      // the mask as a 1-bit sign plane instead of `ref` (ConvArgs::mask_bits)
};

// sign plane requested from a forward launch with the fused activation (ConvArgs::sign_bits); whether the kernel that
// served the launch produced it (only the 3x3 patch tile without split-K does): gg_last_sign_bits_written
struct BitArgs {
  unsigned* sign = nullptr;
  float* amax = nullptr;               // per-image max |output| (ConvArgs::amax_out)
  const unsigned short* xlimb = nullptr;    // operand in limb form + its per-image exponents (ConvArgs::xlimb, xlimb_e)
  const int* xlimb_e = nullptr;
};

constexpr int kNotFusedEarly = GG_NOT_SERVED;

struct ActArgs {
  int on = 0;
  const float* noise = nullptr;
  const float* noise_w = nullptr;
  const float* bias = nullptr;
  float alpha = 0.f, gain = 1.f;
  const float* residual = nullptr;     // ConvArgs::residual (independent of `on`)
};

int conv2d_entry(float* y, const float* x, const float* wmat, const unsigned short* wsplit, long long wsplit_stride,
                 int limbs, const float* in_scale, const float* out_scale, const float* bias, int batch, int groups,
                 int cin_g, int cout_g, int h, int w, int ksize, int stride, int pad, int mode, int out_h, int out_w,
                 void* stream, const ActArgs& act = ActArgs(), const MaskArgs& mask = MaskArgs(),
                 const BitArgs& bits = BitArgs()) {
  g_sign_bits_written = 0;
  g_amax_written = 0;
  if (batch <= 0 || groups <= 0 || cin_g <= 0 || cout_g <= 0) return 0;
  if (!y || !x || (!wmat && !wsplit) || h <= 0 || w <= 0) return gg::fail(-2, "conv2d: bad arguments");
  if (ksize != 1 && ksize != 3) return gg::fail(-2, "conv2d: kernel size %d not supported (1 or 3)", ksize);
  if (groups > 65535) return gg::fail(-2, "conv2d: too many groups");
  if (mode != 0 && mode != 1) return gg::fail(-2, "conv2d: mode must be 0 or 1");
  if (pad < 0) return gg::fail(-2, "conv2d: negative padding");
  if (mode == 0 && (stride < 1 || stride > 2)) return gg::fail(-2, "conv2d: stride must be 1 or 2");
  if (mode == 1 && stride != 2)
    return gg::fail(-2, "conv2d: transposed mode is implemented for stride 2 (stride 1 = mode 0 with flipped taps)");
  // format code: low bits = limb count, bit 4 = binary16 limbs (see Limb<>), bit 5 = the operand x is a GRADIENT (binary16
  // limbs only: the block exponent's E = 0 band starts at 2^5 instead of 2^-3, conv_common.h)
  const bool grad_operand = (limbs & 32) != 0 || mask.ref != nullptr || mask.bits != nullptr;
  limbs &= ~32;
  const bool f16 = (limbs & 16) != 0;
  if (f16 && limbs != 18) return gg::fail(-2, "conv2d_split: binary16 limbs come in pairs (code 18 / 50)");
  limbs &= 15;
  if (limbs) {
    if (limbs < 1 || limbs > 3) return gg::fail(-2, "conv2d_split: limbs must be 1, 2, 3 or 18");
    if (cin_g % BKS != 0) return gg::fail(-2, "conv2d_split: cin per group must be a multiple of %d", BKS);
    if (in_scale && (reinterpret_cast<uintptr_t>(in_scale) & 15)) return gg::fail(-2, "conv2d_split: in_scale must be 16-byte aligned");
    if (reinterpret_cast<uintptr_t>(wsplit) & 15) return gg::fail(-2, "conv2d_split: weights must be 16-byte aligned");
    if (wsplit_stride * 2 * limbs >= (1LL << 31)) return gg::fail(-2, "conv2d_split: weight buffer too large");
  }
  ConvArgs a;
  a.y = y; a.x = x; a.wmat = wmat; a.in_scale = in_scale; a.out_scale = out_scale; a.bias = bias;
  a.wsplit = wsplit; a.wsplit_stride = wsplit_stride;
  a.part = nullptr; a.part_stride = 0;
  a.f16 = f16 ? 1 : 0;
  a.exp_lo = grad_operand ? 32.f : 0.125f;
  a.xlimb = bits.xlimb ? bits.xlimb : g_prelimb;      // (g_prelimb: measurement only, gg_debug_set_prelimb)
  a.xlimb_e = bits.xlimb ? bits.xlimb_e : nullptr;
  g_prelimb = nullptr;
  a.amax_out = (act.on && groups == 1) ? bits.amax : nullptr;
  a.acc_scale = f16 ? 1.f / kF16WeightScale : 1.f;
  a.mask_ref = mask.ref; a.mask_alpha = mask.alpha; a.mask_gain = mask.gain;
  a.mask_bits = mask.bits;
  a.sign_bits = (act.on && groups == 1 && cout_g % 32 == 0) ? bits.sign : nullptr;
  a.bit_words = mask.bits ? cin_g / 32 : cout_g / 32;
  a.act = act.on; a.act_noise = act.noise; a.act_noise_w = act.noise_w; a.act_bias = act.bias;
  a.residual = act.residual;
  // only the generic tiles (and the split-K reduce) add a residual: 1x1 split-precision launches always run on those
  if (a.residual && !(ksize == 1 && limbs && mode == 0 && !act.on)) return kNotFusedEarly;
  a.act_alpha = act.alpha; a.act_gain = act.gain;
  a.batch = batch; a.groups = groups; a.cin_g = cin_g; a.cout_g = cout_g; a.h = h; a.w = w;
  if (mode == 0) {
    a.oh = (h + 2 * pad - ksize) / stride + 1;
    a.ow = (w + 2 * pad - ksize) / stride + 1;
    if ((out_h > 0 && out_h != a.oh) || (out_w > 0 && out_w != a.ow)) return gg::fail(-2, "conv2d: output size mismatch");
  } else {
    a.oh = (h - 1) * stride - 2 * pad + ksize;
    a.ow = (w - 1) * stride - 2 * pad + ksize;
    // output_padding: rows / columns beyond the natural size receive no contribution (zeros)
    if (out_h > 0) { if (out_h < a.oh || out_h >= a.oh + stride) return gg::fail(-2, "conv2d: bad out_h"); a.oh = out_h; }
    if (out_w > 0) { if (out_w < a.ow || out_w >= a.ow + stride) return gg::fail(-2, "conv2d: bad out_w"); a.ow = out_w; }
  }
  if (a.oh <= 0 || a.ow <= 0) return 0;
  static const int nt_env = env_int("GG_NT_STORE", -1);             // measurement override: 0 = never, 1 = always
  a.nt_store = nt_env >= 0 ? nt_env
                           : ((long long)batch * groups * cout_g * a.oh * a.ow * 4 > kNtStoreBytes ? 1 : 0);
  hipStream_t st = gg::as_stream(stream);
  const int rc = ksize == 3 ? conv_dispatch<3>(a, stride, pad, mode, st, limbs) : conv_dispatch<1>(a, stride, pad, mode, st, limbs);
  if (rc != 0) g_sign_bits_written = g_amax_written = 0;
  return rc;
}
}  // namespace

extern "C" int gg_conv2d_f32(float* y, const float* x, const float* wmat, const float* in_scale,
                             const float* out_scale, const float* bias, int batch, int groups, int cin_g, int cout_g,
                             int h, int w, int ksize, int stride, int pad, int mode, int out_h, int out_w,
                             void* stream) {
  return conv2d_entry(y, x, wmat, nullptr, 0, 0, in_scale, out_scale, bias, batch, groups, cin_g, cout_g, h, w, ksize,
                      stride, pad, mode, out_h, out_w, stream);
}

extern "C" int gg_conv2d_split_f32(float* y, const float* x, const unsigned short* wsplit, long long limb_stride,
                                   int limbs, const float* in_scale, const float* out_scale, const float* bias,
                                   int batch, int groups, int cin_g, int cout_g, int h, int w, int ksize, int stride,
                                   int pad, int mode, int out_h, int out_w, void* stream) {
  return conv2d_entry(y, x, nullptr, wsplit, limb_stride, limbs, in_scale, out_scale, bias, batch, groups, cin_g,
                      cout_g, h, w, ksize, stride, pad, mode, out_h, out_w, stream);
}

// gg_conv2d_split_f32 + bias + leaky ReLU for ANY served geometry (ResBlock's blur -> 3x3 / stride-2 convolution ->
// FusedLeakyReLU, networks.py:375-386): the stride-2 patch tile carries the activation in its epilogue, split-K launches in
// their reduce pass, everything else in an in-place pass of the library's own (no separate entry-point call).
extern "C" int gg_conv2d_split_act_f32(float* y, const float* x, const unsigned short* wsplit, long long limb_stride,
                                       int limbs, const float* in_scale, const float* out_scale, const float* act_bias,
                                       float alpha, float gain, int batch, int groups, int cin_g, int cout_g, int h,
                                       int w, int ksize, int stride, int pad, int mode, int out_h, int out_w,
                                       void* stream) {
  if (!wsplit || !(limbs & 15)) return gg::fail(-2, "conv2d_split_act: split weights missing");
  const int oh = mode == 0 ? (h + 2 * pad - ksize) / (stride > 0 ? stride : 1) + 1 : out_h;
  const int ow = mode == 0 ? (w + 2 * pad - ksize) / (stride > 0 ? stride : 1) + 1 : out_w;
  if (((long long)oh * ow) % 4 != 0 || (reinterpret_cast<uintptr_t>(y) & 15)) return kNotFusedEarly;
  ActArgs act;
  act.on = 1; act.bias = act_bias; act.alpha = alpha; act.gain = gain;
  return conv2d_entry(y, x, nullptr, wsplit, limb_stride, limbs, in_scale, out_scale, nullptr, batch, groups, cin_g,
                      cout_g, h, w, ksize, stride, pad, mode, out_h, out_w, stream, act);
}

// gg_conv2d_split_f32 of a 1x1 convolution + residual: y = conv1x1(x) * out_scale + bias + residual, residual of y's
// shape (ResBlock's skip branch with the merge inside, networks.py:386-393).  GG_NOT_SERVED for other kernel sizes.
extern "C" int gg_conv1x1_split_residual_f32(float* y, const float* x, const unsigned short* wsplit,
                                             long long limb_stride, int limbs, const float* in_scale,
                                             const float* out_scale, const float* bias, const float* residual, int batch,
                                             int groups, int cin_g, int cout_g, int h, int w, int stride, void* stream) {
  if (!wsplit || !(limbs & 15) || !residual) return gg::fail(-2, "conv1x1_split_residual: null pointer");
  ActArgs act;
  act.residual = residual;
  return conv2d_entry(y, x, nullptr, wsplit, limb_stride, limbs, in_scale, out_scale, bias, batch, groups, cin_g, cout_g,
                      h, w, 1, stride, 0, 0, 0, 0, stream, act);
}

extern "C" int gg_modconv3x3_act_f32(float* y, const float* x, const float* wmat, const unsigned short* wsplit,
                                     long long limb_stride, int limbs, const float* in_scale,
                                     const float* out_scale, const float* noise, const float* noise_weight,
                                     const float* act_bias, float alpha, float gain, int batch, int cin, int cout,
                                     int h, int w, void* stream) {
  return gg_modconv3x3_act_bits_f32(y, x, wmat, wsplit, limb_stride, limbs, in_scale, out_scale, noise, noise_weight,
                                    act_bias, alpha, gain, batch, cin, cout, h, w, nullptr, stream);
}

extern "C" int gg_modconv3x3_act_bits_f32(float* y, const float* x, const float* wmat, const unsigned short* wsplit,
                                          long long limb_stride, int limbs, const float* in_scale,
                                          const float* out_scale, const float* noise, const float* noise_weight,
                                          const float* act_bias, float alpha, float gain, int batch, int cin,
                                          int cout, int h, int w, unsigned int* sign_bits, void* stream) {
  if (noise && !noise_weight) return gg::fail(-2, "modconv3x3_act: noise without its weight");
  if ((h * w) % 4 != 0 || (reinterpret_cast<uintptr_t>(noise) & 15) || (reinterpret_cast<uintptr_t>(y) & 15))
    return gg::fail(-2, "modconv3x3_act: H*W must be a multiple of 4 and y / noise 16-byte aligned");
  if (limbs == 0 && !wmat) return gg::fail(-2, "modconv3x3_act: fp32 weights missing");
  if (limbs != 0 && !wsplit) return gg::fail(-2, "modconv3x3_act: split weights missing");
  ActArgs act;
  act.on = 1; act.noise = noise; act.noise_w = noise_weight; act.bias = act_bias; act.alpha = alpha; act.gain = gain;
  BitArgs bits;
  bits.sign = sign_bits;
  return conv2d_entry(y, x, limbs ? nullptr : wmat, limbs ? wsplit : nullptr, limb_stride, limbs, in_scale, out_scale,
                      nullptr, batch, 1, cin, cout, h, w, 3, 1, 1, 0, 0, 0, stream, act, MaskArgs(), bits);
}

extern "C" int gg_modconv3x3_act_amax_f32(float* y, const float* x, const float* wmat, const unsigned short* wsplit,
                                          long long limb_stride, int limbs, const float* in_scale,
                                          const float* out_scale, const float* noise, const float* noise_weight,
                                          const float* act_bias, float alpha, float gain, int batch, int cin,
                                          int cout, int h, int w, unsigned int* sign_bits, float* amax_out,
                                          void* stream) {
  if (noise && !noise_weight) return gg::fail(-2, "modconv3x3_act: noise without its weight");
  if ((h * w) % 4 != 0 || (reinterpret_cast<uintptr_t>(noise) & 15) || (reinterpret_cast<uintptr_t>(y) & 15))
    return gg::fail(-2, "modconv3x3_act: H*W must be a multiple of 4 and y / noise 16-byte aligned");
  if (limbs == 0 && !wmat) return gg::fail(-2, "modconv3x3_act: fp32 weights missing");
  if (limbs != 0 && !wsplit) return gg::fail(-2, "modconv3x3_act: split weights missing");
  ActArgs act;
  act.on = 1; act.noise = noise; act.noise_w = noise_weight; act.bias = act_bias; act.alpha = alpha; act.gain = gain;
  BitArgs bits;
  bits.sign = sign_bits;
  bits.amax = amax_out;
  return conv2d_entry(y, x, limbs ? nullptr : wmat, limbs ? wsplit : nullptr, limb_stride, limbs, in_scale, out_scale,
                      nullptr, batch, 1, cin, cout, h, w, 3, 1, 1, 0, 0, 0, stream, act, MaskArgs(), bits);
}

// Transposed 3x3 / stride 2 convolution of an operand that arrives in limb form (gg_torgb_limb_f32): GG_NOT_SERVED, and
// nothing launched, unless the 16-channel-chunk tile with 64 output channels per block serves the shape.
extern "C" int gg_convT3x3s2_prelimb_f32(float* y, const unsigned short* xlimb, const int* xlimb_exp,
                                         const unsigned short* wsplit, long long limb_stride, const float* out_scale,
                                         const float* bias, int batch, int cin, int cout, int h, int w, int pad,
                                         int out_h, int out_w, void* stream) {
  if (!xlimb || !xlimb_exp || !wsplit) return gg::fail(-2, "convT3x3s2_prelimb: null pointer");
  if ((reinterpret_cast<uintptr_t>(xlimb) & 15)) return gg::fail(-2, "convT3x3s2_prelimb: xlimb must be 16-byte aligned");
  if (cin % BKS != 0) return kNotFused;
  BitArgs bits;
  bits.xlimb = xlimb;
  bits.xlimb_e = xlimb_exp;
  g_prelimb_served = 0;
  // x is never read on this path; the entry's null check wants a pointer
  const int rc = conv2d_entry(y, reinterpret_cast<const float*>(xlimb), nullptr, wsplit, limb_stride, 18, nullptr,
                              out_scale, bias, batch, 1, cin, cout, h, w, 3, 2, pad, 1, out_h, out_w, stream, ActArgs(),
                              MaskArgs(), bits);
  if (rc != 0) return rc;
  return g_prelimb_served ? 0 : gg::fail(-5, "convT3x3s2_prelimb: the launch was not served by the limb-form tile");
}

// ToRGB (modulated 1x1 convolution to 3 channels, no demodulation, + bias: networks.py:352-372) of y AND y's limb form for
// the next resolution's up-sampling convolution (style `next_style`, per-image exponent into xlimb_exp), one read of y.
extern "C" int gg_torgb_limb_f32(float* rgb, unsigned short* xlimb, int* xlimb_exp, const float* y,
                                 const float* rgb_wmat, const float* rgb_style, const float* rgb_bias,
                                 const float* next_style, const float* amax, int batch, int cin, long long hw,
                                 void* stream) {
  if (batch <= 0) return 0;
  if (!rgb || !xlimb || !xlimb_exp || !y || !rgb_wmat || !rgb_style || !next_style || !amax)
    return gg::fail(-2, "torgb_limb: null pointer");
  if ((reinterpret_cast<uintptr_t>(y) & 15) || (reinterpret_cast<uintptr_t>(rgb) & 15) ||
      (reinterpret_cast<uintptr_t>(xlimb) & 15))
    return gg::fail(-2, "torgb_limb: y, rgb and xlimb must be 16-byte aligned");
  if (t16_torgb_limb(rgb, xlimb, xlimb_exp, y, rgb_wmat, rgb_style, rgb_bias, next_style, amax, batch, cin, hw,
                     gg::as_stream(stream)) != 0)
    return kNotFused;
  return gg::launch_status("torgb_limb");
}

extern "C" int gg_conv3x3_masked_dgrad_f32(float* y, const float* x, const float* mask_ref, float alpha, float gain,
                                           const unsigned short* wsplit, long long limb_stride, int limbs,
                                           const float* in_scale, const float* out_scale, int batch, int cin,
                                           int cout, int h, int w, void* stream) {
  if (!mask_ref) return gg::fail(-2, "conv3x3_masked_dgrad: mask_ref missing");
  if (limbs != 1 && limbs != 2 && limbs != 18) return kNotFused;
  MaskArgs mask;
  mask.ref = mask_ref; mask.alpha = alpha; mask.gain = gain;
  return conv2d_entry(y, x, nullptr, wsplit, limb_stride, limbs, in_scale, out_scale, nullptr, batch, 1, cin, cout, h,
                      w, 3, 1, 1, 0, 0, 0, stream, ActArgs(), mask);
}

extern "C" int gg_conv3x3_masked_dgrad_bits_f32(float* y, const float* x, const unsigned int* mask_bits, float alpha,
                                                float gain, const unsigned short* wsplit, long long limb_stride,
                                                int limbs, const float* in_scale, const float* out_scale, int batch,
                                                int cin, int cout, int h, int w, void* stream) {
  if (!mask_bits) return gg::fail(-2, "conv3x3_masked_dgrad_bits: mask_bits missing");
  if (limbs != 18 || cin % 32 != 0) return kNotFused;      // the bit-plane gather exists on the binary16-limb tiles
  MaskArgs mask;
  mask.bits = mask_bits; mask.alpha = alpha; mask.gain = gain;
  return conv2d_entry(y, x, nullptr, wsplit, limb_stride, limbs, in_scale, out_scale, nullptr, batch, 1, cin, cout, h,
                      w, 3, 1, 1, 0, 0, 0, stream, ActArgs(), mask);
}

// Masked data gradient of a 3x3 conv + (leaky) ReLU layer with <= 4 INPUT channels (the perceptual trunk's RGB stem):
// dx (N, cout <= 4, H, W) = conv3x3(dy * lrelu'(y), wmat), lrelu'(y) from the sign plane.  wmat: fp32 GEMM layout
// [(ci, ky, kx)][co] of the DATA-GRADIENT convolution (gg_conv_pack_weight_f32 with transpose_io = flip = 1).
extern "C" int gg_conv3x3_fewout_masked_bits_f32(float* dx, const float* dy, const unsigned int* mask_bits, float alpha,
                                                 float gain, const float* wmat, int batch, int cin, int cout, int h,
                                                 int w, void* stream) {
  if (batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0) return 0;
  if (!dx || !dy || !mask_bits || !wmat) return gg::fail(-2, "conv3x3_fewout_masked_bits: null pointer");
  ConvArgs a = {};
  a.y = dx; a.x = dy; a.wmat = wmat;
  a.batch = batch; a.groups = 1; a.cin_g = cin; a.cout_g = cout; a.h = h; a.w = w; a.oh = h; a.ow = w;
  if (cin % 32 != 0 || !fewout3_serves(a, 1, 1, 0)) return kNotFused;
  a.mask_bits = mask_bits; a.bit_words = cin / 32; a.mask_alpha = alpha; a.mask_gain = gain;
  g_sign_bits_written = 0;
  g_amax_written = 0;
  return launch_conv3x3_fewout_masked(a, gg::as_stream(stream));
}

extern "C" int gg_conv_pack_weights_many(const void* jobs, int njobs, void* stream) {
  if (njobs <= 0) return 0;
  if (!jobs || njobs > 65535) return gg::fail(-2, "conv_pack_weights_many: bad arguments");
  pack_weight_many_kernel<<<dim3(PACK_MANY_BLOCKS, (unsigned)njobs), 256, 0, gg::as_stream(stream)>>>(
      reinterpret_cast<const PackJob*>(jobs));
  return gg::launch_status("conv_pack_weights_many");
}

extern "C" int gg_conv_pack_weight_split(unsigned short* wsplit, const float* w, int groups, int cout_g, int cin_g,
                                         int kh, int kw, int transpose_io, int flip, float scale, int limbs,
                                         void* stream) {
  const long long total = (long long)groups * cout_g * cin_g * kh * kw;
  if (total <= 0) return 0;
  if (!wsplit || !w || (limbs & 15) < 1 || (limbs & 15) > 3 || (limbs & ~31) || ((limbs & 16) && (limbs & 15) != 2))
    return gg::fail(-2, "conv_pack_weight_split: limbs must be 1, 2, 3 (bf16) or 18 (two binary16 limbs)");
  pack_weight_split_kernel<<<gg::stream_grid(total, 256), 256, 0, gg::as_stream(stream)>>>(
      wsplit, w, total, total, cout_g, cin_g, kh, kw, transpose_io, flip, scale, limbs);
  return gg::launch_status("conv_pack_weight_split");
}

namespace {
int wgrad_entry(float* dw, const float* x, const float* dy, int batch, int groups, int cin_g, int cout_g, int h,
                int w, int ksize, int stride, int pad, float scale, int limbs, void* stream, bool accumulate = false,
                float* workspace = nullptr, long long workspace_bytes = 0, const float* mask_ref = nullptr,
                float mask_alpha = 0.f, float mask_gain = 1.f, float* dbias = nullptr) {
  if (groups <= 0 || cin_g <= 0 || cout_g <= 0) return 0;
  if (!dw || !x || !dy) return gg::fail(-2, "conv2d_wgrad: null pointer");
  if (ksize != 1 && ksize != 3) return gg::fail(-2, "conv2d_wgrad: kernel size must be 1 or 3");
  if (stride < 1 || groups > 65535) return gg::fail(-2, "conv2d_wgrad: bad arguments");
  hipStream_t st = gg::as_stream(stream);
  WgradArgs a;
  a.dw = dw; a.x = x; a.dy = dy;
  a.batch = batch; a.groups = groups; a.cin_g = cin_g; a.cout_g = cout_g; a.h = h; a.w = w;
  a.oh = (h + 2 * pad - ksize) / stride + 1;
  a.ow = (w + 2 * pad - ksize) / stride + 1;
  a.stride = stride; a.pad = pad; a.scale = scale;
  a.mask_ref = mask_ref; a.mask_alpha = mask_alpha; a.mask_gain = mask_gain; a.dbias = dbias;
  a.part = nullptr; a.accumulate = accumulate ? 1 : 0;
  a.jtot = cin_g * ksize * ksize;
  auto zero_dw = [&]() -> int {  // the kernels combine their K-splits with atomic adds: start from zero unless adding
    if (accumulate) return 0;
    hipError_t e = hipMemsetAsync(dw, 0, sizeof(float) * (size_t)groups * cout_g * a.jtot, st);
    return e == hipSuccess ? 0 : gg::fail((int)e, "conv2d_wgrad: memset failed");
  };
  if (batch <= 0 || a.oh <= 0 || a.ow <= 0) return zero_dw();
  if (limbs) {
    if (limbs < 1 || limbs > 3) return gg::fail(-2, "conv2d_wgrad_split: limbs must be 1, 2 or 3");
    if ((a.oh * a.ow) % BKS != 0 || a.ow % 4 != 0 || (reinterpret_cast<uintptr_t>(dy) & 15))
      return gg::fail(-2, "conv2d_wgrad_split: needs OH*OW %% 32 == 0, OW %% 4 == 0 and 16-byte aligned dy");
  }
  if (ksize == 1 && stride == 1 && pad == 0 && groups == 1 && cin_g <= 4 && !mask_ref && ((long long)h * w) % 64 == 0) {
    // few-input-channel 1x1 stem: streaming reduction (fp32 exact in every precision mode)
    const int nblocks = 2 * gg::kNumCu;
    const long long need = (long long)nblocks * cout_g * cin_g * (long long)sizeof(float);
    if (!workspace || workspace_bytes < need) {             // no caller workspace: the stream's scratch
      workspace = reinterpret_cast<float*>(gg::scratch(st, (size_t)need));
      if (!workspace) return -3;
      workspace_bytes = need;
    }
    if (workspace && workspace_bytes >= need) {
      dim3 grid((unsigned)nblocks, (unsigned)((cout_g + 63) / 64));
      const long long hw = (long long)h * w;
      switch (cin_g) {
        case 1: wgrad_1x1_smallcin_kernel<1><<<grid, 256, 0, st>>>(workspace, x, dy, batch, cout_g, hw); break;
        case 2: wgrad_1x1_smallcin_kernel<2><<<grid, 256, 0, st>>>(workspace, x, dy, batch, cout_g, hw); break;
        case 3: wgrad_1x1_smallcin_kernel<3><<<grid, 256, 0, st>>>(workspace, x, dy, batch, cout_g, hw); break;
        default: wgrad_1x1_smallcin_kernel<4><<<grid, 256, 0, st>>>(workspace, x, dy, batch, cout_g, hw); break;
      }
      int rc = gg::launch_status("wgrad_1x1_smallcin");
      if (rc) return rc;
      const int count = cout_g * cin_g;
      partial_sum_kernel<<<(count + 3) / 4, 256, 0, st>>>(dw, workspace, nblocks, count, scale, accumulate ? 1 : 0);
      return gg::launch_status("partial_sum");
    }
  }
  static const bool no_tiny_wgrad = getenv("GG_NO_TINY_WGRAD") != nullptr;          // measurement switch
  // (1x1 only: 62 -> 27 us on the 512 -> 512 skip at 4^2.  The 3x3 instantiation - 144 LDS-fed FMAs per thread and sample -
  // measured 81 us against the generic tile's 78 on the two 3x3 layers at 4^2 and is not dispatched: GG_TINY_WGRAD3=1 tries it.)
  static const bool tiny3 = getenv("GG_TINY_WGRAD3") != nullptr;
  if (!no_tiny_wgrad && groups == 1 && !mask_ref && a.oh * a.ow <= TS_MAXP && h * w <= TS_MAXHW && (a.oh * a.ow) % BKS != 0 &&
      (ksize == 1 || tiny3) && (cout_g + TS_CO - 1) / TS_CO <= 65535 && (cin_g + TS_CI - 1) / TS_CI <= 65535) {
    // tiny outputs (4 x 4): the split-precision slab loaders do not apply; see wgrad_tiny_spatial_kernel
    NOTE_KERNEL("wgrad_tiny_spatial<k%d>", ksize);
    dim3 grid((unsigned)((cout_g + TS_CO - 1) / TS_CO), (unsigned)((cin_g + TS_CI - 1) / TS_CI));
    if (ksize == 3)
      wgrad_tiny_spatial_kernel<3><<<grid, 256, 0, st>>>(dw, x, dy, batch, cin_g, cout_g, h, w, a.oh, a.ow, stride, pad, scale,
                                                         accumulate ? 1 : 0);
    else
      wgrad_tiny_spatial_kernel<1><<<grid, 256, 0, st>>>(dw, x, dy, batch, cin_g, cout_g, h, w, a.oh, a.ow, stride, pad, scale,
                                                         accumulate ? 1 : 0);
    return gg::launch_status("wgrad_tiny_spatial");
  }
  static const bool no_fewout_wgrad = getenv("GG_NO_FEWOUT_WGRAD") != nullptr;      // measurement switch
  if (!no_fewout_wgrad && ksize == 3 && stride == 1 && pad == 1 && groups == 1 && cout_g <= 4 && !mask_ref && cin_g <= 65535 &&
      (long long)batch * h * w < (1LL << 31)) {
    // few-output-channel layer (the flow head's 512 -> 2): streaming reduction, exact fp32 in every precision mode
    NOTE_KERNEL("wgrad3x3_fewout");
    switch (cout_g) {
      case 1: wgrad3x3_fewout_kernel<1><<<cin_g, 256, 0, st>>>(dw, x, dy, batch, cin_g, h, w, scale, accumulate ? 1 : 0); break;
      case 2: wgrad3x3_fewout_kernel<2><<<cin_g, 256, 0, st>>>(dw, x, dy, batch, cin_g, h, w, scale, accumulate ? 1 : 0); break;
      case 3: wgrad3x3_fewout_kernel<3><<<cin_g, 256, 0, st>>>(dw, x, dy, batch, cin_g, h, w, scale, accumulate ? 1 : 0); break;
      default: wgrad3x3_fewout_kernel<4><<<cin_g, 256, 0, st>>>(dw, x, dy, batch, cin_g, h, w, scale, accumulate ? 1 : 0); break;
    }
    return gg::launch_status("wgrad3x3_fewout");
  }
  const bool strip16 = a.w == 16 && a.h % 2 == 0 && !mask_ref;            // 16-wide images: two rows per slab
  if (limbs && ksize == 3 && stride == 1 && pad == 1 && (a.w % 32 == 0 || strip16) &&
      (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    // row-streaming kernel: (128 co x 32 ci) tiles, or (64 x 64) for narrow outputs
    const bool narrow = cout_g <= 64;
    const int tco = narrow ? 64 : 128, tci = narrow ? 64 : 32;
    a.tiles_co = (cout_g + tco - 1) / tco;
    a.tiles_j = (cin_g + tci - 1) / tci;
    const int segs = strip16 ? 1 : a.w / 32;
    const long long tiles = (long long)a.tiles_co * a.tiles_j * groups;
    const long long units = (long long)batch * segs;
    // two resident blocks per CU; more blocks only add partial tiles to write and reduce
    long long rblocks = (2LL * gg::kNumCu + tiles * units - 1) / (tiles * units);
    const long long max_rb = (a.h + 7) / 8;                             // >= 8 rows per block
    if (rblocks > max_rb) rblocks = max_rb;
    if (rblocks < 1) rblocks = 1;
    int rows_per_block = (int)((a.h + rblocks - 1) / rblocks);
    if (strip16) rows_per_block += rows_per_block & 1;
    rblocks = (a.h + rows_per_block - 1) / rows_per_block;
    const long long total_units = units * rblocks;
    // chain K-units per block when there are more blocks than two per CU (small images: a unit is a few slabs)
    long long upb = tiles * total_units / (2LL * gg::kNumCu);
    if (upb < 1) upb = 1;
    if (upb > 8) upb = 8;
    // every split costs a partial tile of 144 KB per (co, ci) tile: bound the partials at 1 GiB by chaining more
    // K-units per block (fewer, longer blocks) instead of asking the allocator for whatever the shape implies
    constexpr long long kMaxPartialBytes = 1LL << 30;
    while (upb < total_units &&
           tiles * ((total_units + upb - 1) / upb) * 9LL * 4096 * (long long)sizeof(float) > kMaxPartialBytes)
      upb *= 2;
    const long long splits = (total_units + upb - 1) / upb;
    const long long need_dw = tiles * splits * 9LL * 4096 * (long long)sizeof(float);
    const long long need = need_dw + (dbias ? splits * (long long)groups * cout_g * (long long)sizeof(float) : 0);
    if ((!workspace || workspace_bytes < need) && need < (2LL << 30)) {   // no caller workspace: the stream's scratch
      workspace = reinterpret_cast<float*>(gg::scratch(st, (size_t)need));
      if (!workspace) return -3;
      workspace_bytes = need;
    }
    float* dbws = dbias ? workspace + need_dw / sizeof(float) : nullptr;
    if (workspace && workspace_bytes >= need && splits <= 65535 && (long long)a.tiles_co * a.tiles_j < (1LL << 31)) {
      dim3 grid((unsigned)(a.tiles_co * a.tiles_j), (unsigned)splits, (unsigned)groups);
      if (strip16) {
        if (limbs <= 2) {
          if (narrow) LIMBS12(limbs, conv3x3_wgrad_rows_kernel<L, 64, 64, false, 16><<<grid, 256, 0, st>>>(a, segs, (int)rblocks, rows_per_block, (int)upb, workspace, dbws));
          else LIMBS12(limbs, conv3x3_wgrad_rows_kernel<L, 128, 32, false, 16><<<grid, 256, 0, st>>>(a, segs, (int)rblocks, rows_per_block, (int)upb, workspace, dbws));
        } else {
          if (narrow) conv3x3_wgrad_rows_kernel<3, 64, 64, false, 16><<<grid, 256, 0, st>>>(a, segs, (int)rblocks, rows_per_block, (int)upb, workspace, dbws);
          else conv3x3_wgrad_rows_kernel<3, 128, 32, false, 16><<<grid, 256, 0, st>>>(a, segs, (int)rblocks, rows_per_block, (int)upb, workspace, dbws);
        }
      } else if (a.mask_ref) {   // limbs == 2 (checked by the entry point)
        if (narrow) LIMBS12(limbs, conv3x3_wgrad_rows_kernel<L, 64, 64, true><<<grid, 256, 0, st>>>(a, segs, (int)rblocks, rows_per_block, (int)upb, workspace, dbws));
        else LIMBS12(limbs, conv3x3_wgrad_rows_kernel<L, 128, 32, true><<<grid, 256, 0, st>>>(a, segs, (int)rblocks, rows_per_block, (int)upb, workspace, dbws));
      } else if (limbs <= 2) {
        if (narrow) LIMBS12(limbs, conv3x3_wgrad_rows_kernel<L, 64, 64><<<grid, 256, 0, st>>>(a, segs, (int)rblocks, rows_per_block, (int)upb, workspace, dbws));
        else LIMBS12(limbs, conv3x3_wgrad_rows_kernel<L, 128, 32><<<grid, 256, 0, st>>>(a, segs, (int)rblocks, rows_per_block, (int)upb, workspace, dbws));
      } else {
        if (narrow) conv3x3_wgrad_rows_kernel<3, 64, 64><<<grid, 256, 0, st>>>(a, segs, (int)rblocks, rows_per_block, (int)upb, workspace, dbws);
        else conv3x3_wgrad_rows_kernel<3, 128, 32><<<grid, 256, 0, st>>>(a, segs, (int)rblocks, rows_per_block, (int)upb, workspace, dbws);
      }
      int rc = gg::launch_status("conv3x3_wgrad_rows");
      if (rc) return rc;
      const long long total = tiles * 9LL * 4096;
      if (splits >= 64)
        wgrad_reduce_kernel<8><<<gg::stream_grid(total * 8, 256), 256, 0, st>>>(
            dw, workspace, groups, cout_g, cin_g, a.tiles_co, a.tiles_j, tco, tci, (int)splits, scale, accumulate ? 1 : 0,
            dbias, dbws);
      else
        wgrad_reduce_kernel<1><<<gg::stream_grid(total, 256), 256, 0, st>>>(
            dw, workspace, groups, cout_g, cin_g, a.tiles_co, a.tiles_j, tco, tci, (int)splits, scale, accumulate ? 1 : 0,
            dbias, dbws);
      return gg::launch_status("wgrad_reduce");
    }
  }
  static const bool s2_rows = env_int("GG_S2_WGRAD", 1) != 0;           // measurement switch: 0 = the generic kernel
  if (s2_rows && limbs && ksize == 3 && stride == 2 && pad == 0 && !mask_ref && a.ow >= 16 &&
      (long long)cin_g * h * w * 4 < (1LL << 31)) {
    // row-streaming stride-2 kernel (conv_s2_wgrad.hip): the plan of the stride-1 kernel over the OUTPUT rows / columns
    const bool narrow = cout_g <= 64;
    const int tco = narrow ? 64 : 128, tci = narrow ? 64 : 32;
    a.tiles_co = (cout_g + tco - 1) / tco;
    a.tiles_j = (cin_g + tci - 1) / tci;
    const int segs = (a.ow + 31) / 32;
    const long long tiles = (long long)a.tiles_co * a.tiles_j * groups;
    const long long units = (long long)batch * segs;
    long long rblocks = (2LL * gg::kNumCu + tiles * units - 1) / (tiles * units);
    const long long max_rb = (a.oh + 7) / 8;                            // >= 8 rows per block
    if (rblocks > max_rb) rblocks = max_rb;
    if (rblocks < 1) rblocks = 1;
    const int rows_per_block = (int)((a.oh + rblocks - 1) / rblocks);
    rblocks = (a.oh + rows_per_block - 1) / rows_per_block;
    const long long total_units = units * rblocks;
    long long upb = tiles * total_units / (2LL * gg::kNumCu);
    if (upb < 1) upb = 1;
    if (upb > 8) upb = 8;
    constexpr long long kMaxPartialBytes = 1LL << 30;
    while (upb < total_units &&
           tiles * ((total_units + upb - 1) / upb) * 9LL * 4096 * (long long)sizeof(float) > kMaxPartialBytes)
      upb *= 2;
    const long long splits = (total_units + upb - 1) / upb;
    const long long need = tiles * splits * 9LL * 4096 * (long long)sizeof(float);
    if ((!workspace || workspace_bytes < need) && need < (2LL << 30)) {
      workspace = reinterpret_cast<float*>(gg::scratch(st, (size_t)need));
      if (!workspace) return -3;
      workspace_bytes = need;
    }
    if (workspace && workspace_bytes >= need && splits <= 65535 && (long long)a.tiles_co * a.tiles_j < (1LL << 31)) {
      dim3 grid((unsigned)(a.tiles_co * a.tiles_j), (unsigned)splits, (unsigned)groups);
      s2_wgrad_rows_launch(a, limbs, narrow, segs, (int)rblocks, rows_per_block, (int)upb, workspace, grid, st);
      int rc = gg::launch_status("conv3x3s2_wgrad_rows");
      if (rc) return rc;
      const long long total = tiles * 9LL * 4096;
      if (splits >= 64)
        wgrad_reduce_kernel<8><<<gg::stream_grid(total * 8, 256), 256, 0, st>>>(
            dw, workspace, groups, cout_g, cin_g, a.tiles_co, a.tiles_j, tco, tci, (int)splits, scale, accumulate ? 1 : 0,
            nullptr, nullptr);
      else
        wgrad_reduce_kernel<1><<<gg::stream_grid(total, 256), 256, 0, st>>>(
            dw, workspace, groups, cout_g, cin_g, a.tiles_co, a.tiles_j, tco, tci, (int)splits, scale, accumulate ? 1 : 0,
            nullptr, nullptr);
      return gg::launch_status("wgrad_reduce");
    }
  }
  if (mask_ref) return kNotFused;          // only the row-streaming kernel applies the mask; nothing was launched
  a.ktot = (long long)batch * a.oh * a.ow;
  a.tiles_co = (cout_g + WT - 1) / WT;
  a.tiles_j = (a.jtot + WT - 1) / WT;
  const long long tiles = (long long)a.tiles_co * a.tiles_j * groups;
  const int slab = limbs ? BKS : WBK;
  long long splits = (4LL * gg::kNumCu + tiles - 1) / tiles;
  const long long max_splits = (a.ktot + 8 * slab - 1) / (8 * slab);       // >= 8 slabs per split
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  if (splits > 65535) splits = 65535;
  long long kps = (a.ktot + splits - 1) / splits;
  kps = (kps + slab - 1) / slab * slab;
  splits = (a.ktot + kps - 1) / kps;
  a.k_per_split = kps;
  const long long count = (long long)groups * cout_g * a.jtot;
  a.part = nullptr;
  a.accumulate = accumulate ? 1 : 0;
  if (splits > 1) {
    a.part = reinterpret_cast<float*>(gg::scratch(st, sizeof(float) * (size_t)count * splits));
    if (!a.part) return -3;
  }
  dim3 grid((unsigned)(a.tiles_co * a.tiles_j), (unsigned)splits, (unsigned)groups);
  if (limbs == 1 || limbs == 2) {
    if (ksize == 3) LIMBS12(limbs, conv_wgrad_split_kernel<3, L><<<grid, 256, 0, st>>>(a));
    else LIMBS12(limbs, conv_wgrad_split_kernel<1, L><<<grid, 256, 0, st>>>(a));
  } else if (limbs == 3) {
    if (ksize == 3) conv_wgrad_split_kernel<3, 3><<<grid, 256, 0, st>>>(a);
    else conv_wgrad_split_kernel<1, 3><<<grid, 256, 0, st>>>(a);
  } else if (ksize == 3) {
    conv_wgrad_kernel<3><<<grid, 256, 0, st>>>(a);
  } else {
    conv_wgrad_kernel<1><<<grid, 256, 0, st>>>(a);
  }
  int rc = gg::launch_status("conv2d_wgrad");
  if (rc || !a.part) return rc;
  if (splits >= 32 && count <= 16384)           // many partials of few elements: one wave per element (otherwise a
                                                // thread per element: coalesced across elements, 512 x 512 1x1 weights
                                                // took 28 us with 262144 waves reading 4-byte pieces)
    partial_sum_kernel<<<(unsigned)((count + 3) / 4), 256, 0, st>>>(dw, a.part, (int)splits, (int)count, scale,
                                                                    accumulate ? 1 : 0);
  else
    partial_sum_flat_kernel<<<gg::stream_grid(count, 256), 256, 0, st>>>(dw, a.part, (int)splits, count, scale,
                                                                         accumulate ? 1 : 0);
  return gg::launch_status("wgrad_partial_sum");
}
}  // namespace

extern "C" int gg_conv2d_wgrad_f32(float* dw, const float* x, const float* dy, int batch, int groups, int cin_g,
                                   int cout_g, int h, int w, int ksize, int stride, int pad, float scale,
                                   void* stream) {
  return wgrad_entry(dw, x, dy, batch, groups, cin_g, cout_g, h, w, ksize, stride, pad, scale, 0, stream);
}

extern "C" int gg_conv2d_wgrad_split_f32(float* dw, const float* x, const float* dy, int batch, int groups,
                                         int cin_g, int cout_g, int h, int w, int ksize, int stride, int pad,
                                         float scale, int limbs, void* stream) {
  return wgrad_entry(dw, x, dy, batch, groups, cin_g, cout_g, h, w, ksize, stride, pad, scale, limbs, stream);
}

extern "C" int gg_conv2d_wgrad_acc_f32(float* dw, const float* x, const float* dy, int batch, int groups, int cin_g,
                                       int cout_g, int h, int w, int ksize, int stride, int pad, float scale,
                                       int limbs, void* stream) {
  return wgrad_entry(dw, x, dy, batch, groups, cin_g, cout_g, h, w, ksize, stride, pad, scale, limbs, stream, true);
}

extern "C" int gg_conv2d_wgrad_ws_f32(float* dw, const float* x, const float* dy, int batch, int groups, int cin_g,
                                      int cout_g, int h, int w, int ksize, int stride, int pad, float scale,
                                      int limbs, int accumulate, float* workspace, long long workspace_bytes,
                                      void* stream) {
  return wgrad_entry(dw, x, dy, batch, groups, cin_g, cout_g, h, w, ksize, stride, pad, scale, limbs, stream,
                     accumulate != 0, workspace, workspace_bytes);
}

extern "C" int gg_conv3x3_masked_wgrad_f32(float* dw, float* dbias, const float* x, const float* dy,
                                           const float* mask_ref, float alpha, float gain, int batch, int cin,
                                           int cout, int h, int w, float scale, int limbs, int accumulate,
                                           float* workspace, long long workspace_bytes, void* stream) {
  if (!mask_ref) return gg::fail(-2, "conv3x3_masked_wgrad: mask_ref missing");
  if ((limbs != 1 && limbs != 2) || (reinterpret_cast<uintptr_t>(mask_ref) & 15)) return kNotFused;
  return wgrad_entry(dw, x, dy, batch, 1, cin, cout, h, w, 3, 1, 1, scale, limbs, stream, accumulate != 0, workspace,
                     workspace_bytes, mask_ref, alpha, gain, dbias);
}

extern "C" int gg_plane_dot_f32(float* out, const float* a, const float* b, int planes, long long hw, void* stream) {
  if (planes <= 0) return 0;
  if (!out || !a || !b || hw < 0) return gg::fail(-2, "plane_dot: bad arguments");
  plane_dot_kernel<<<planes, 256, 0, gg::as_stream(stream)>>>(out, a, b, hw);
  return gg::launch_status("plane_dot");
}

// MEASUREMENT ONLY (scripts/prelimb_probe.py): gg_debug_limb_convert writes the style-scaled
// two-binary16-limb, channel-fastest form of an activation; gg_debug_set_prelimb hands it to the NEXT convolution launch
// of this thread, where the transposed 16-channel-chunk tile (64 co, binary16) stages it with wide loads instead of
// converting fp32 in its loader.  Not declared in the public header: nothing in the product path calls them.
extern "C" int gg_debug_limb_convert(unsigned short* out, const float* x, const float* in_scale, int planes,
                                     long long hw, void* stream) {
  if (!out || !x || planes <= 0 || planes % 16 != 0 || hw <= 0 || hw >= (1LL << 31))
    return gg::fail(-2, "debug_limb_convert: bad arguments");
  t16_limb_convert(out, x, in_scale, planes, (int)hw, gg::as_stream(stream));
  return gg::launch_status("debug_limb_convert");
}
extern "C" int gg_debug_set_prelimb(const unsigned short* xlimb) {
  g_prelimb = xlimb;
  return 0;
}

// Diagnostic: resident workgroups per CU of the main convolution kernels on the current device, as
// "name=blocks;..." (what hipOccupancyMaxActiveBlocksPerMultiprocessor reports).
extern "C" int gg_debug_conv_occupancy(char* out, int out_len) {
  if (!out || out_len <= 0) return gg::fail(-2, "debug_conv_occupancy: no buffer");
  struct Entry { const char* name; const void* fn; int threads; };
  const Entry entries[] = {
      {"conv3x3_patch<2,true,256>", reinterpret_cast<const void*>(&conv3x3_patch_kernel<2, true, 256>), 512},
      {"conv3x3_patch<2,false,128>", reinterpret_cast<const void*>(&conv3x3_patch_kernel<2, false, 128>), 256},
      {"conv3x3_patch<3,true,128>", reinterpret_cast<const void*>(&conv3x3_patch_kernel<3, true, 128>), 256},
      {"convT3x3s2_patch<2,true,128>", reinterpret_cast<const void*>(&convT3x3s2_patch_kernel<2, true, 128>), 512},
      {"convT3x3s2_patch<2,true,64>", reinterpret_cast<const void*>(&convT3x3s2_patch_kernel<2, true, 64>), 256},
      {"conv_split<3,0,2,true,256>", reinterpret_cast<const void*>(&conv_split_kernel<3, 0, 2, true, 256>), 512},
      {"conv_split<3,0,2,false,128>", reinterpret_cast<const void*>(&conv_split_kernel<3, 0, 2, false, 128>), 256},
      {"conv_wgrad_split<3,2>", reinterpret_cast<const void*>(&conv_wgrad_split_kernel<3, 2>), 256},
      {"conv_igemm<3,0,2,2,2,2,true>", reinterpret_cast<const void*>(&conv_igemm_kernel<3, 0, 2, 2, 2, 2, true>), 256},
  };
  int pos = 0;
  for (const Entry& e : entries) {
    int blocks = -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, e.fn, e.threads, 0) != hipSuccess) blocks = -1;
    pos += snprintf(out + pos, pos < out_len ? out_len - pos : 0, "%s=%d;", e.name, blocks);
    if (pos >= out_len) break;
  }
  return 0;
}
