// a3 / a4  3x3 / stride 2 TRANSPOSED convolution on 16-channel chunks (round 5).
//
// Who runs it: StyleGAN2's up-sampling ModulatedConv2d (reference models/stylegan2/networks.py:254-265:
// conv_transpose2d(stride 2, padding 0) of the (N, Cin, H, W) activation, (2H+1)^2 out, then Blur) and the data
// gradient of every 3x3 / stride-2 convolution of the STN (networks.py:455-480).
//   out[co, 2q + p - pad] = sum_{ci, j} x[ci, q - j] * W[co, ci, k = p + 2j],   p in {0,1}, k < 3
//
// convT3x3s2_patch_kernel (conv_mfma.hip) runs the same all-classes-in-one-pass formulation on 32-channel chunks in
// 80-byte LDS rows, one 8-wave block per CU.  Its counters (profiles/r04_q_transposed_tile_sq_counters.txt): matrix
// pipe busy in 34 % of the cycles, 19 % of the LDS cycles bank conflicts, 1.27 LDS instructions per MFMA, and - the
// round-3 attribution - 19 % of a launch spent in the output stores, 10 % in the rest of the epilogue, 17 % in the
// activation loads, all ADDITIVE because a block that owns its CU has nobody to compute under its memory phases.
//
// This kernel keeps the formulation (a wave owns 32 co x 64 q and all four parity classes: 128 accumulator registers;
// tap t accumulates into the set of its class (ky & 1, kx & 1)) and changes what surrounds it:
//   * a block is 64 co x 128 q on FOUR waves, two blocks per CU: the epilogue / prologue of one block runs under the
//     other block's tap loop, a layer has twice as many (half as long) blocks - the partial last round costs half as
//     much - and Cout = 64 layers (STN 128 -> 64 gradient, C4's 128 -> 64 @256^2 -> 513^2) multiply no zero rows.
//     (The 8-wave 128 co form is instantiated too: TCO = 128.)
//   * 16-channel chunks in UNPADDED 32-byte LDS rows, 16-byte halves XOR-swizzled with bit 3 of the row index
//     (conv_s2_patch.hip's layout: a 16-lane group of a ds_read_b128 reads 16 consecutive rows -> 16 distinct slots of
//     the 256-byte bank row).  Patch 2 limbs x 258 rows x 32 B = 16.5 KB, ALL NINE taps of the chunk 9 x 2 x 64 x 32 B
//     = 36.9 KB: 53 KB per block.  One barrier interval per chunk: 54 MFMAs per wave between two barriers.
//   * taps ordered by the patch offset (1 - (ky >> 1), 1 - (kx >> 1)) they read: the four taps with ky, kx <= 1 (one
//     per parity class) share their patch fragments, two pairs share theirs - 4 patch-fragment sets per chunk instead
//     of 9; 34 ds_read_b128 per 54 MFMAs.
//   * the next chunk's 17 activation loads and 9 weight loads are issued two / one per tap inside the tap loop.
// Tile geometry (interior tiles TH x TW of the (H+1) x (W+1) q-grid, TW a power of two; the extra column in "edge"
// tiles of shape 128 x 1), limb formats, block exponent, split-K partials and the class-interleaving epilogue are
// those of convT3x3s2_patch_kernel.
#include "conv_common.h"

namespace {

using namespace gg_conv;

constexpr int T_CH = 16;              // input channels per chunk = K of one MFMA step
constexpr int T_RB = 32;              // bytes per LDS row
constexpr int T_TQ = 128;             // q positions per block

// order index -> tap (ky * 3 + kx): {(0,0) (0,1) (1,0) (1,1)} {(0,2) (1,2)} {(2,0) (2,1)} {(2,2)}
__device__ __forceinline__ constexpr int t16_tap(int i) {
  return i == 0 ? 0 : i == 1 ? 1 : i == 2 ? 3 : i == 3 ? 4 : i == 4 ? 2 : i == 5 ? 5 : i;
}
__device__ __forceinline__ constexpr int t16_group(int i) { return i < 4 ? 0 : i < 6 ? 1 : i < 8 ? 2 : 3; }

// PRELIMB (measurement only): the activation arrives as a.xlimb - already style-scaled and split into
// two binary16 limbs, channel-fastest in the 32-byte rows this kernel keeps in LDS - so the loader is four 16-byte loads
// and four ds_write_b128 per patch pixel and chunk: no style multiply, no amax, no v_cvt, no block exponent (E = 0).
template <bool IN_SCALE, int TCO, bool F16, bool PRELIMB = false>
__global__ __launch_bounds__(TCO * 4, 2) void convT3x3s2_c16_kernel(const ConvArgs a, int tw_log2_in, int tiles_y,
                                                                    int edge_tiles, int pad) {
  using L = Limb<F16>;
  constexpr int LIMBS = 2, NJ = 2, TQ = T_TQ, NT = TCO * 4, NW = NT / 64, PWAVES = TQ / 64;
  constexpr int CPT = 16 * 256 / NT;                  // patch channels per thread (thread = patch pixel x channel part)
  constexpr int PATCH_MAX = 2 * TQ + 2;
  constexpr int P_BYTES = PATCH_MAX * T_RB;           // one limb of the patch
  constexpr int W_BYTES = TCO * T_RB;                 // one (tap, limb) weight slab
  constexpr int MAIN_BYTES = LIMBS * P_BYTES + 9 * LIMBS * W_BYTES;
  constexpr int STAGE_BYTES = NW * 8 * 128 * 4, EPI_BYTES = 2 * TCO * 4;
  constexpr int SMEM_BYTES = MAIN_BYTES > STAGE_BYTES + EPI_BYTES ? MAIN_BYTES : STAGE_BYTES + EPI_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES + 64];
  float* sAmax = reinterpret_cast<float*>(smem + SMEM_BYTES);      // per-wave operand maxima (BlockExp)
  unsigned char* sP = smem;                                         // [limb][patch pixel][32 B]
  unsigned char* sW = smem + LIMBS * P_BYTES;                       // [tap][limb][co][32 B]

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

  // ---- patch gather: thread -> (patch pixel pp < 256, CPT of the chunk's 16 channels); the patch origin is (y0-1, x0-1)
  const int pp = tid & 255, cpart = tid >> 8;          // channels [cpart * CPT, cpart * CPT + CPT)
  const bool pin = pp < PP;
  unsigned pvoff;
  {
    const int pr = pp / PW, pc = pp - pr * PW;
    const int iy = y0 + pr - 1, ix = x0 + pc - 1;
    const bool pok = pin & ((unsigned)iy < (unsigned)a.h) & ((unsigned)ix < (unsigned)a.w);
    pvoff = pok ? (unsigned)(cpart * CPT * hw + iy * a.w + ix) * 4u : kOobOffset;
  }
  // PRELIMB: the same patch pixel, as a byte offset into the limb-form tensor (64 B per pixel and chunk)
  const __amdgpu_buffer_rsrc_t lr = uniform_rsrc(reinterpret_cast<const float*>(PRELIMB ? a.xlimb : nullptr) +
                                                 (PRELIMB ? (size_t)chan0 * hw : 0), a.cin_g * hw * 4);
  unsigned plimb = kOobOffset, llimb = kOobOffset;
  if (PRELIMB) {
    const int pr = pp / PW, pc = pp - pr * PW;
    const int iy = y0 + pr - 1, ix = x0 + pc - 1;
    if (pin && (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w) plimb = (unsigned)(iy * a.w + ix) * 64u;
    const int l2 = 256 + (tid & 1);                      // pixels 256 / 257: threads 0..7 take one 16-byte piece each
    const int qr = l2 / PW, qc = l2 - qr * PW;
    const int jy = y0 + qr - 1, jx = x0 + qc - 1;
    if (tid < 8 && l2 < PP && (unsigned)jy < (unsigned)a.h && (unsigned)jx < (unsigned)a.w)
      llimb = (unsigned)(jy * a.w + jx) * 64u + (unsigned)((tid >> 1) & 3) * 16u;
  }
  U4 xq[4], xlq = U4{0u, 0u, 0u, 0u};
  const int lpp = 256 + (tid & 1), lci = (tid >> 1) & (T_CH - 1);      // patch pixels 256, 257 (128 x 1 and 2 x 128 tiles)
  const bool lin = tid < 2 * T_CH && lpp < PP;
  unsigned lvoff;
  {
    const int pr = lpp / PW, pc = lpp - pr * PW;
    const int iy = y0 + pr - 1, ix = x0 + pc - 1;
    const bool lok = lin & ((unsigned)iy < (unsigned)a.h) & ((unsigned)ix < (unsigned)a.w);
    lvoff = lok ? (unsigned)(lci * hw + iy * a.w + ix) * 4u : kOobOffset;
  }
  // ---- weight rows: thread -> (limb, co row, 8-channel part) of every tap: nine 16-byte pieces per chunk
  const int wpart = tid & 1, wrow = (tid >> 1) & (TCO - 1), wsel = tid / (2 * TCO);
  const bool w_ok = (co0 + wrow) < a.cout_g;
  const int kfull = 9 * a.cin_g;
  const __amdgpu_buffer_rsrc_t wr = uniform_rsrc(reinterpret_cast<const float*>(a.wsplit),
                                                 (int)(a.wsplit_stride * 2 * LIMBS));
  const unsigned wvoff =
      w_ok ? (unsigned)((((size_t)g * a.cout_g + co0 + wrow) * kfull + wpart * 8) * 2 + (size_t)wsel * a.wsplit_stride * 2)
           : kOobOffset;
  const int wlds = (wsel * TCO + wrow) * T_RB + (((wpart ^ (wrow >> 3)) & 1) << 4);

  const int chunk0 = split * a.slabs_per_split;
  int chunk1 = chunk0 + a.slabs_per_split;
  if (chunk1 > a.nslabs) chunk1 = a.nslabs;

  float xa[CPT], xl = 0.f;
  U4 wv[9];

  // slice i (0..8) of the next chunk's loads: two of the lane's patch channels and the weight piece of order-tap i
  auto load_slice = [&](int chunk, int i) {
    const int cbase = __builtin_amdgcn_readfirstlane(chunk * T_CH * hw * 4);
    if (PRELIMB) {
      if (i < 4) xq[i] = buffer_load_u4(lr, plimb, cbase + i * 16);       // (chunk * hw * 64 B == cbase)
      if (i == 4) xlq = buffer_load_u4(lr, llimb, cbase);
    } else {
#pragma unroll
      for (int j = 0; j < CPT; ++j)
        if (j >= 2 * i && j < 2 * i + 2) xa[j] = buffer_load_f32(xr, pvoff, cbase + j * hw * 4);
      if (i == 8) xl = buffer_load_f32(xr, lvoff, cbase);
    }
    const int soff = __builtin_amdgcn_readfirstlane((t16_tap(i) * a.cin_g + chunk * T_CH) * 2);
    wv[i] = buffer_load_u4(wr, wvoff, soff);
  };
  // the chunk's registers in their final fp32 form (style); binary16 limbs: + this wave's largest magnitude
  BlockExp bexp;
  if (PRELIMB && a.xlimb_e) bexp.e = a.xlimb_e[pn];          // the producer pass scaled the whole image by 2^-E
  auto prep_patch = [&](int chunk) {
    if (PRELIMB) return;
    if (IN_SCALE) {
      if (pin) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) xa[j] *= sg[chunk * T_CH + cpart * CPT + j];
      }
      if (lin) xl *= sg[chunk * T_CH + lci];
    }
    if (F16) {
      float m = fabsf(xl);
#pragma unroll
      for (int j = 0; j < CPT; ++j) m = fmaxf(m, fabsf(xa[j]));
      publish_wave_amax(m, sAmax, wid, lane);
    }
  };
  auto store_patch = [&]() {
    if (PRELIMB) {
      if (pin) {
#pragma unroll
        for (int l = 0; l < LIMBS; ++l)
#pragma unroll
          for (int q = 0; q < 2; ++q)
            *reinterpret_cast<U4*>(sP + l * P_BYTES + pp * T_RB + (((q ^ (pp >> 3)) & 1) << 4)) = xq[l * 2 + q];
      }
      if (tid < 8 && (256 + (tid & 1)) < PP) {
        const int l2 = 256 + (tid & 1), piece = (tid >> 1) & 3;
        *reinterpret_cast<U4*>(sP + (piece >> 1) * P_BYTES + l2 * T_RB + ((((piece & 1) ^ (l2 >> 3)) & 1) << 4)) = xlq;
      }
      return;
    }
    if (F16 && bexp.e != 0) {                  // a uniform branch: most tiles never leave E = 0
      const float ps = exp2i(-bexp.e);
#pragma unroll
      for (int j = 0; j < CPT; ++j) xa[j] *= ps;
      xl *= ps;
    }
    if (pin) {
#pragma unroll
      for (int l = 0; l < LIMBS; ++l) {
#pragma unroll
        for (int q = 0; q < CPT / 8; ++q) {
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
          const int half = cpart * (CPT / 8) + q;          // which 8 channels of the row
          *reinterpret_cast<U4*>(sP + l * P_BYTES + pp * T_RB + (((half ^ (pp >> 3)) & 1) << 4)) =
              U4{pk[0], pk[1], pk[2], pk[3]};
        }
      }
    }
    if (lin) {
      float v = xl;
#pragma unroll
      for (int l = 0; l < LIMBS; ++l) {
        const unsigned short hb = L::one(v, l == 0);
        *reinterpret_cast<unsigned short*>(sP + l * P_BYTES + lpp * T_RB + ((((lci >> 3) ^ (lpp >> 3)) & 1) << 4) +
                                           (lci & 7) * 2) = hb;
        v -= L::back(hb);
      }
    }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int t = 0; t < 9; ++t) *reinterpret_cast<U4*>(sW + t16_tap(t) * LIMBS * W_BYTES + wlds) = wv[t];
  };

  f32x16 acc[4][NJ];                        // [parity class py*2+px][pixel sub-tile]
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][j][r] = 0.f;

  const int kh = lane >> 5, l31 = lane & 31;
  // byte offsets of the lane's B fragments: [patch offset group][sub-tile]; group 0: (dy, dx) = (1, 1), 1: (1, 0),
  // 2: (0, 1), 3: (0, 0)   (x[q - j]: patch row / column (q - origin) + 1 - j)
  int boff[4][NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int p = (wpix * NJ + j) * 32 + l31;
    const int r = p >> tw_log2, c = p & (TW - 1);
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      const int dy = gi < 2 ? 1 : 0, dx = (gi & 1) ? 0 : 1;
      const int row = (r + dy) * PW + c + dx;
      boff[gi][j] = row * T_RB + (((kh ^ (row >> 3)) & 1) << 4);
    }
  }
  const int aoff = (wco * 32 + l31) * T_RB + (((kh ^ (l31 >> 3)) & 1) << 4);

  if (chunk0 < chunk1) {
#pragma unroll
    for (int i = 0; i < 9; ++i) load_slice(chunk0, i);
    if (F16 && !PRELIMB) prep_patch(chunk0);             // published by the chunk loop's first barrier
    for (int chunk = chunk0; chunk < chunk1; ++chunk) {
      __syncthreads();                       // the previous chunk's readers are done with sP / sW
      if (F16 && !PRELIMB) {          // block exponent of this chunk (rescales the accumulators if it grew)
        const float f = block_exp_update(bexp, read_block_amax<NW>(sAmax), a.exp_lo);
        if (f != 1.f) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[c][j][r] *= f;
        }
      } else if (!PRELIMB) {
        prep_patch(chunk);
      }
      store_patch();
      store_w();
      __syncthreads();
      const bool more = chunk + 1 < chunk1;
      // (the last chunk re-loads itself: unconditional loads keep the tap loop one basic block)
      const int nchunk = more ? chunk + 1 : chunk;
      // the nine taps in patch-offset order; the weight fragments of tap i + 1 are fetched before the MFMAs of tap i, a
      // group's patch fragments behind the last MFMA of the previous group (the co-resident block covers that round trip)
      bf16x8 fa[2][LIMBS], fb[LIMBS][NJ];
      auto fetch_a = [&](int slot, int i) {
#pragma unroll
        for (int l = 0; l < LIMBS; ++l)
          fa[slot][l] = *reinterpret_cast<const bf16x8*>(sW + (t16_tap(i) * LIMBS + l) * W_BYTES + aoff);
      };
      auto fetch_b = [&](int gi) {
#pragma unroll
        for (int l = 0; l < LIMBS; ++l)
#pragma unroll
          for (int j = 0; j < NJ; ++j) fb[l][j] = *reinterpret_cast<const bf16x8*>(sP + l * P_BYTES + boff[gi][j]);
      };
      fetch_a(0, 0);
      fetch_b(0);
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const int slot = i & 1;
        if (i + 1 < 9) fetch_a(slot ^ 1, i + 1);
        load_slice(nchunk, i);
        __builtin_amdgcn_sched_barrier(0);
        const int t = t16_tap(i);
        const int ky = t / 3, kx = t - ky * 3;
        const int cls = (ky & 1) * 2 + (kx & 1);
#pragma unroll
        for (int sum = LIMBS - 1; sum >= 0; --sum)         // smallest terms first
#pragma unroll
          for (int la = 0; la <= sum; ++la) {
            const int lb = sum - la;
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[cls][j] = L::mfma(fa[slot][la], fb[lb][j], acc[cls][j]);
          }
        __builtin_amdgcn_sched_barrier(0);
        if (i + 1 < 9 && t16_group(i + 1) != t16_group(i)) fetch_b(t16_group(i + 1));
      }
      // binary16 limbs: the next chunk's registers landed during the nine taps; published by the loop's top barrier
      if (F16 && !PRELIMB && more) prep_patch(chunk + 1);
    }
  }
  const float esc = F16 ? exp2i(bexp.e) : 1.f;          // undo the block exponent (exact)

  const int ohw = a.oh * a.ow;
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
  // Interleave the classes through LDS (convT3x3s2_patch_kernel's epilogue): per pass 8 channels x (2 * 64) outputs of
  // one output-row parity; each wave owns its staging rows (wave-level fences only); the two column classes of a q
  // position are adjacent in the output row: 8-byte LDS writes, 16-byte reads, 16-byte dword-aligned buffer stores.
  __syncthreads();                                          // sP / sW are dead from here on
  float* stage = reinterpret_cast<float*>(smem) + wid * (8 * 128);
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
          const unsigned off = (unsigned)(co * ohw + oy * a.ow + ox) * 4u;
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

}  // namespace

namespace gg_conv {

bool t16_serves(const ConvArgs& a) {
  return a.cin_g % T_CH == 0 && (long long)a.cin_g * a.h * a.w * 4 < (1LL << 31) &&
         (long long)a.cout_g * a.oh * a.ow * 4 < (1LL << 31);
}

#define T16_LAUNCH(SC, TCO, F) convT3x3s2_c16_kernel<SC, TCO, F><<<grid, TCO * 4, 0, st>>>(a, tw_log2, tiles_y, edge, pad)
#define T16_LAUNCH_F(SC, TCO) do { if (a.f16) T16_LAUNCH(SC, TCO, true); else T16_LAUNCH(SC, TCO, false); } while (0)
// fp32 NCHW x (x style) -> the limb form of ConvArgs::xlimb (measurement only: what a producer epilogue would write)
__global__ __launch_bounds__(256) void limb_convert_kernel(unsigned short* __restrict__ out, const float* __restrict__ x,
                                                           const float* __restrict__ scale, int planes16, int hw) {
  const long long total = (long long)planes16 * hw;          // (image x chunk) x pixel
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int pc = (int)(i / hw), p = (int)(i - (long long)pc * hw);
    const float* src = x + ((size_t)pc * 16) * hw + p;
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = src[(size_t)j * hw] * (scale ? scale[(size_t)pc * 16 + j] : 1.f);
    unsigned pk[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      pk[0][e] = Limb<true>::pack2(v[2 * e], v[2 * e + 1], true);
      pk[1][e] = Limb<true>::pack2(v[2 * e] - Limb<true>::lo(pk[0][e]), v[2 * e + 1] - Limb<true>::hi(pk[0][e]), false);
    }
    U4* dst = reinterpret_cast<U4*>(out + (size_t)i * 32);
    dst[0] = U4{pk[0][0], pk[0][1], pk[0][2], pk[0][3]};
    dst[1] = U4{pk[0][4], pk[0][5], pk[0][6], pk[0][7]};
    dst[2] = U4{pk[1][0], pk[1][1], pk[1][2], pk[1][3]};
    dst[3] = U4{pk[1][4], pk[1][5], pk[1][6], pk[1][7]};
  }
}

// ---- round 6: ToRGB + limb form in ONE pass over the activation ----------------------------------------------------------
// The output y of a resolution's second StyledConv has two readers: the ToRGB layer (modulated 1x1, 3 outputs: a streaming
// reduction over the channels, conv1x1_fewout_kernel) and the next resolution's up-sampling convolution (the transposed
// tile above).  This kernel is the former with the latter's operand as a by-product: while a wave holds 16 channels x 4
// pixels of y for the RGB dot products, it also multiplies them by the UP-CONVOLUTION's style, splits them into two
// binary16 limbs and stores them channel-fastest, 64 bytes per pixel and chunk - the LDS rows of the PRELIMB loader.  y is
// read once (as before); the limb form costs its write.
// Exponent: one per image, E[n] = f16_block_exp(amax_y[n] * max_c |style[n][c]|) (>= the true largest operand: the band
// rule of conv_common.h on an upper bound); amax_y[n] comes from the producing convolution's epilogue (ConvArgs::amax_out).
// Block = 256 pixels of one image (lane: 4 consecutive pixels); the four waves split the channels into four contiguous
// ranges and meet in LDS - the summation order of conv1x1_fewout_kernel, so the RGB output is bitwise that kernel's.
constexpr int TL_MAX_CIN = 1024;
template <int NCO>
__global__ __launch_bounds__(256) void torgb_limb_kernel(float* __restrict__ rgb, unsigned short* __restrict__ xlimb,
                                                         int* __restrict__ xexp, const float* __restrict__ x,
                                                         const float* __restrict__ wmat,
                                                         const float* __restrict__ rgb_style,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ next_style,
                                                         const float* __restrict__ amax, int cin, long long hw) {
  __shared__ float sw[NCO][TL_MAX_CIN];
  __shared__ float ss[TL_MAX_CIN];
  __shared__ float4 red[4][NCO][64];
  __shared__ float smax[4];
  const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, g = tid >> 6;
  float m = 0.f;
  for (int i = tid; i < cin; i += 256) {
    const float sv = next_style[(size_t)n * cin + i];
    ss[i] = sv;
    m = fmaxf(m, fabsf(sv));
  }
  for (int i = tid; i < cin * NCO; i += 256) {
    const int ci = i / NCO, j = i - ci * NCO;
    sw[j][ci] = wmat[i] * rgb_style[(size_t)n * cin + ci];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if (lane == 0) smax[g] = m;
  __syncthreads();
  const float bound = amax[n] * fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
  const int e = bound > 0.f ? f16_block_exp(bound, 0.125f) : 0;
  if (blockIdx.x == 0 && tid == 0) xexp[n] = e;
  const float ps = exp2i(-e);
  const long long p = (long long)blockIdx.x * 256 + lane * 4;
  const bool ok = p < hw;                                   // hw % 4 == 0: a lane's four pixels are in or out together
  float4 acc[NCO];
#pragma unroll
  for (int j = 0; j < NCO; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int nchunks = cin / T_CH;
  if (ok) {
    const int cpw = (nchunks + 3) / 4, c0 = g * cpw, c1 = (c0 + cpw < nchunks) ? c0 + cpw : nchunks;
    for (int chunk = c0; chunk < c1; ++chunk) {
      const float* src = x + ((size_t)n * cin + (size_t)chunk * T_CH) * hw + p;
      float4 v[T_CH];
#pragma unroll
      for (int k = 0; k < T_CH; ++k) v[k] = *reinterpret_cast<const float4*>(src + (size_t)k * hw);
#pragma unroll
      for (int k = 0; k < T_CH; ++k) {
#pragma unroll
        for (int j = 0; j < NCO; ++j) {
          const float wj = sw[j][chunk * T_CH + k];
          acc[j].x += v[k].x * wj; acc[j].y += v[k].y * wj; acc[j].z += v[k].z * wj; acc[j].w += v[k].w * wj;
        }
      }
      // the limb form: exactly the arithmetic of the fp32 loader (style multiply, power-of-two scale when E != 0, leading
      // limb rounded toward zero, residual to nearest)
#pragma unroll
      for (int k = 0; k < T_CH; ++k) {
        const float sv = ss[chunk * T_CH + k];
        v[k].x *= sv; v[k].y *= sv; v[k].z *= sv; v[k].w *= sv;
      }
      if (e != 0) {
#pragma unroll
        for (int k = 0; k < T_CH; ++k) { v[k].x *= ps; v[k].y *= ps; v[k].z *= ps; v[k].w *= ps; }
      }
      U4* dst = reinterpret_cast<U4*>(xlimb + (((size_t)n * nchunks + chunk) * hw + p) * 32);
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        unsigned l0[8], l1[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float a0 = px == 0 ? v[2 * q].x : px == 1 ? v[2 * q].y : px == 2 ? v[2 * q].z : v[2 * q].w;
          const float a1 = px == 0 ? v[2 * q + 1].x : px == 1 ? v[2 * q + 1].y : px == 2 ? v[2 * q + 1].z : v[2 * q + 1].w;
          l0[q] = Limb<true>::pack2(a0, a1, true);
          l1[q] = Limb<true>::pack2(a0 - Limb<true>::lo(l0[q]), a1 - Limb<true>::hi(l0[q]), false);
        }
        dst[px * 4 + 0] = U4{l0[0], l0[1], l0[2], l0[3]};
        dst[px * 4 + 1] = U4{l0[4], l0[5], l0[6], l0[7]};
        dst[px * 4 + 2] = U4{l1[0], l1[1], l1[2], l1[3]};
        dst[px * 4 + 3] = U4{l1[4], l1[5], l1[6], l1[7]};
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
    const float bi = bias ? bias[j] : 0.f;
    r.x += bi; r.y += bi; r.z += bi; r.w += bi;
    *reinterpret_cast<float4*>(rgb + ((size_t)n * NCO + j) * hw + p) = r;
  }
}

int t16_torgb_limb(float* rgb, unsigned short* xlimb, int* xexp, const float* x, const float* wmat,
                   const float* rgb_style, const float* bias, const float* next_style, const float* amax, int batch,
                   int cin, long long hw, hipStream_t st) {
  if (cin % T_CH != 0 || cin > TL_MAX_CIN || hw % 4 != 0 || batch > 65535) return -1;
  dim3 grid((unsigned)((hw + 255) / 256), (unsigned)batch);
  torgb_limb_kernel<3><<<grid, 256, 0, st>>>(rgb, xlimb, xexp, x, wmat, rgb_style, bias, next_style, amax, cin, hw);
  return 0;
}

void t16_limb_convert(unsigned short* out, const float* x, const float* scale, int planes, int hw, hipStream_t st) {
  const long long total = (long long)(planes / 16) * hw;
  limb_convert_kernel<<<gg::stream_grid(total, 256), 256, 0, st>>>(out, x, scale, planes / 16, hw);
}

void t16_launch(const ConvArgs& a, int tco, int tw_log2, int tiles_y, int edge, int pad, dim3 grid, hipStream_t st) {
  const bool sc = a.in_scale != nullptr;
  if (a.xlimb && tco == 64 && a.f16) {          // measurement only
    convT3x3s2_c16_kernel<false, 64, true, true><<<grid, 256, 0, st>>>(a, tw_log2, tiles_y, edge, pad);
    return;
  }
  if (tco == 64) {
    if (sc) T16_LAUNCH_F(true, 64);
    else T16_LAUNCH_F(false, 64);
  } else {
    if (sc) T16_LAUNCH_F(true, 128);
    else T16_LAUNCH_F(false, 128);
  }
}

}  // namespace gg_conv
