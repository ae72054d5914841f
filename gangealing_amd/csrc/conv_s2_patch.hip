// a3  3x3 / stride 2 / pad 0 correlation with input-patch reuse (round 4).
//
// Who runs it: the data gradients of the generator's up-convolutions (networks.py:268-272: conv_transpose2d(stride 2)
// -> its adjoint is a stride-2 correlation of dy, (2H+1)^2 -> H^2) and the forward of the STN's down-sampling
// ConvLayers (networks.py:455-480: Blur, then a stride-2 convolution without padding).  conv_split_kernel serves these
// shapes by re-gathering - and re-splitting into limbs - every input element once per tap that touches it: 9 gathers
// per output pixel where only (2 TH + 1)(2 TW + 1) / (TH TW) = 4.6 distinct input pixels exist, ~300 VALU / VMEM
// instructions per 24 MFMAs per wave: instruction-issue bound at 180 - 220 TF/s (profiles/r04_f_conv_layers.txt).
//
// This kernel stages the input patch of a TH x 32 output tile ONCE per 16-channel chunk - (2 TH + 1) x 65 pixels,
// style-scaled, block-exponent-scaled and split into two limbs on the way in - and lets the nine taps read their B
// fragments from it.  What made that fit (round 2 costed the idea with 80-byte rows and dropped it):
//   * 16-channel chunks in UNPADDED 32-byte rows (16 x 2 B), XOR-swizzled in 16-byte halves: half' = half ^ bit 3 of
//     the row index.  ds_read_b128 services 16-lane groups whose rows cover all 16 residues mod 16, so the 16 reads of
//     a group land on 16 distinct 16-byte slots of the 256-byte bank row (row r and r + 8 take opposite halves).
//     Patch of a 4 x 32 tile: 2 limbs x 594 rows x 32 B = 38 KB (80-byte rows of 32 channels: 95 KB).
//   * column-parity planes: output column c, tap kx reads input column 2c + kx = plane (kx & 1), plane column
//     c + (kx >> 1) - the 32 lanes of a fragment read 32 CONSECUTIVE rows of one plane, as in the stride-1 tile.
//   * a whole row of taps (ky fixed, kx = 0..2) per barrier interval: 3 x 2 limbs x 128 rows x 32 B = 24 KB of weights,
//     36 MFMAs per wave between barriers.  62 KB in all: two 4-wave blocks per CU, which fill each other's staging
//     phases (the arrangement that serves the 128-pixel stride-1 tile).
// Per chunk and wave: 40 gathered elements per lane and 108 MFMAs (conv_split_kernel: 72 and 108, plus its per-slab
// address / bounds / exponent work nine times instead of once).
//
// GEMM orientation, limb formats, block exponents, epilogue: as conv3x3_patch_kernel (conv_mfma.hip).
#include "conv_common.h"

namespace {

using namespace gg_conv;

constexpr int S2_CH = 16;             // input channels per chunk = K of one MFMA step
constexpr int S2_RB = 32;             // bytes per LDS row
constexpr int S2_TW = 32;             // output tile width

// S = stride: 2 = the stride-2 / pad-0 correlation described above; 1 = the same tile for 3x3 / stride 1 / pad 1 (one
// plane, a (TH + 2) x 34 patch: 1.59 input pixels per output where the 2 x 64 tile of conv3x3_patch_kernel's 128-pixel
// variant stages 2.06, 42 KB of LDS: three blocks per CU) - see launch_conv_s2_patch's notes for where it is used.
template <int S, bool IN_SCALE, int TPIX, bool F16>
__global__ __launch_bounds__(TPIX * 2, TPIX == 128 ? (S == 1 ? 3 : 2) : 1) void conv3x3s2_patch_kernel(const ConvArgs a) {
  using L = Limb<F16>;
  constexpr int LIMBS = 2, MI = 2, NJ = 2, TCO = 128, NT = TPIX * 2, PWAVES = TPIX / 64, TPI = 3;
  constexpr int S2_PLW = S == 2 ? S2_TW + 1 : S2_TW + 2;     // row pitch of a plane (stride 2: even plane 33 columns, odd 32)
  constexpr int S2_PC = S == 2 ? 2 * S2_TW + 1 : S2_TW + 2;  // patch columns
  constexpr int PAD = S == 2 ? 0 : 1;
  constexpr int TH = TPIX / S2_TW, PH = S == 2 ? 2 * TH + 1 : TH + 2;
  constexpr int NPIX = PH * S2_PC, NITEMS = 2 * NPIX, ROUNDS = (NITEMS + NT - 1) / NT;
  constexpr int PLROWS = PH * S2_PLW, PLANE_BYTES = PLROWS * S2_RB, LIMB_BYTES = S * PLANE_BYTES;
  constexpr int W_BYTES = TCO * S2_RB;                                   // one (tap, limb) weight slab
  constexpr int MAIN_BYTES = LIMBS * LIMB_BYTES + TPI * LIMBS * W_BYTES;
  constexpr int STAGE_BYTES = (NT / 64) * 32 * 64 * 4, EPI_BYTES = (3 * TCO + TPIX) * 4;
  constexpr int SMEM_BYTES = MAIN_BYTES > STAGE_BYTES + EPI_BYTES ? MAIN_BYTES : STAGE_BYTES + EPI_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES + 64 + kS2MaxCin * 4];
  float* sAmax = reinterpret_cast<float*>(smem + SMEM_BYTES);           // per-wave operand maxima (BlockExp)
  float* sStyle = reinterpret_cast<float*>(smem + SMEM_BYTES + 64);     // in_scale of this image-group
  unsigned char* sP = smem;                                             // [limb][plane][row][32 B]
  unsigned char* sW = smem + LIMBS * LIMB_BYTES;                        // [tap of the row][limb][co][32 B]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wco = wid / PWAVES, wpix = wid % PWAVES;
  const unsigned ntiles = (unsigned)a.tiles_co * a.tiles_pix;
  const unsigned logical = gg::xcd_remap(blockIdx.x, ntiles);
  const int tile_co = logical % a.tiles_co, tile_pix = logical / a.tiles_co;
  const int split = blockIdx.y, g = blockIdx.z;
  const int co0 = tile_co * TCO;
  const int hw = a.h * a.w, ohw = a.oh * a.ow;
  const int tiles_x = a.ow / S2_TW, tiles_y = a.oh / TH;
  const int pn = tile_pix / (tiles_x * tiles_y);
  const int trem = tile_pix - pn * tiles_x * tiles_y;
  const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
  const int y0 = ty * TH, x0 = tx * S2_TW;

  const int chan0 = (pn * a.groups + g) * a.cin_g;
  const __amdgpu_buffer_rsrc_t xr = uniform_rsrc(a.x + (size_t)chan0 * hw, a.cin_g * hw * 4);

  // ---- patch gather.  Item = (channel half, patch pixel) in that order: a wave's 64 lanes take 64 consecutive patch
  // pixels (rows of 65 contiguous input pixels) of ONE channel per load; every item is 8 channels = one 16-byte LDS
  // store per limb.  ROUNDS items per thread; the half of an item is lane data (the two halves meet inside one wave).
  unsigned gvo[ROUNDS];          // byte offset of the item's first channel inside the image-group (or out of range)
  int lo[ROUNDS];                // byte offset of its 16 bytes inside a limb's patch; < 0: no such item
  unsigned halfmask = 0;
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int item = r * NT + tid;
    const bool ok = item < NITEMS;
    const int half = item >= NPIX ? 1 : 0;
    const int pix = item - half * NPIX;
    const int pr = pix / S2_PC, pc = pix - pr * S2_PC;
    const int iy = S * y0 + pr - PAD, ix = S * x0 + pc - PAD;
    const bool inb = ok & ((unsigned)iy < (unsigned)a.h) & ((unsigned)ix < (unsigned)a.w);
    gvo[r] = inb ? (unsigned)(half * 8 * hw + iy * a.w + ix) * 4u : kOobOffset;
    const int plane = S == 2 ? (pc & 1) : 0, row = pr * S2_PLW + (S == 2 ? (pc >> 1) : pc);
    lo[r] = ok ? (plane * PLROWS + row) * S2_RB + (((half ^ (row >> 3) ^ plane) & 1) << 4) : -1;
    halfmask |= (unsigned)half << r;
  }
  // ---- weight rows: thread -> (co row, 8-channel part); the first 256 threads move a slab
  const int wrow = (tid >> 1) & (TCO - 1), wpart = tid & 1;
  const bool w_thr = tid < 2 * TCO;
  const bool w_ok = (co0 + wrow) < a.cout_g;
  const int kfull = 9 * a.cin_g;
  const __amdgpu_buffer_rsrc_t wr = uniform_rsrc(reinterpret_cast<const float*>(a.wsplit),
                                                 (int)(a.wsplit_stride * 2 * LIMBS));
  const unsigned wvoff = w_ok ? (unsigned)((((size_t)g * a.cout_g + co0 + wrow) * kfull + wpart * 8) * 2) : kOobOffset;
  const int wlimb = __builtin_amdgcn_readfirstlane((int)(a.wsplit_stride * 2));     // bytes between limb planes
  const int wlds = wrow * S2_RB + (((wpart ^ (wrow >> 3)) & 1) << 4);

  const int chunk0 = split * a.slabs_per_split;
  int chunk1 = chunk0 + a.slabs_per_split;
  if (chunk1 > a.nslabs) chunk1 = a.nslabs;

  if (IN_SCALE) {
    const float* sg = a.in_scale + chan0;
    for (int c = tid; c < a.cin_g; c += NT) sStyle[c] = sg[c];
    __syncthreads();
  }

  float xa[ROUNDS][8];
  U4 wv[TPI][LIMBS];

  auto load_patch = [&](int chunk) {
    const int cbase = __builtin_amdgcn_readfirstlane(chunk * S2_CH * hw * 4);
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
#pragma unroll
      for (int j = 0; j < 8; ++j) xa[r][j] = buffer_load_f32(xr, gvo[r], cbase + j * hw * 4);
  };
  // the chunk's registers in their final fp32 form (style); binary16 limbs: + this wave's largest magnitude
  auto prep_patch = [&](int chunk) {
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      if (IN_SCALE) {
        const float4* s4 = reinterpret_cast<const float4*>(sStyle + chunk * S2_CH + ((halfmask >> r) & 1u) * 8);
        const float4 s0 = s4[0], s1 = s4[1];
        xa[r][0] *= s0.x; xa[r][1] *= s0.y; xa[r][2] *= s0.z; xa[r][3] *= s0.w;
        xa[r][4] *= s1.x; xa[r][5] *= s1.y; xa[r][6] *= s1.z; xa[r][7] *= s1.w;
      }
      if (F16) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) m = fmaxf(fmaxf(m, fabsf(xa[r][j])), fabsf(xa[r][j + 1]));
      }
    }
    if (F16) publish_wave_amax(m, sAmax, wid, lane);
  };
  BlockExp bexp;
  unsigned pk[ROUNDS][LIMBS][4];       // the chunk's items as packed limbs
  auto split_round = [&](int r) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = xa[r][j];
#pragma unroll
    for (int l = 0; l < LIMBS; ++l) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pk[r][l][e] = L::pack2(v[2 * e], v[2 * e + 1], l == 0);
        if (l + 1 < LIMBS) {
          v[2 * e] -= L::lo(pk[r][l][e]);
          v[2 * e + 1] -= L::hi(pk[r][l][e]);
        }
      }
    }
  };
  auto split_patch = [&]() {
    if (F16 && bexp.e != 0) {                     // a uniform BRANCH (most tiles never leave E = 0): as a select the scaling
      const float ps = exp2i(-bexp.e);            // cost a multiply and a v_cndmask per element and chunk (SQ_INSTS_VALU: -20 %)
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) xa[r][j] *= ps;
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) split_round(r);
  };
  auto write_patch = [&]() {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      if (lo[r] >= 0) {
#pragma unroll
        for (int l = 0; l < LIMBS; ++l)
          *reinterpret_cast<U4*>(sP + l * LIMB_BYTES + lo[r]) = U4{pk[r][l][0], pk[r][l][1], pk[r][l][2], pk[r][l][3]};
      }
    }
  };
  // interval iv of a chunk = the taps (ky = iv, kx = 0..2)
  auto load_w = [&](int chunk, int iv) {
    if (!w_thr) return;
#pragma unroll
    for (int u = 0; u < TPI; ++u) {
      const int soff = __builtin_amdgcn_readfirstlane(((iv * TPI + u) * a.cin_g + chunk * S2_CH) * 2);
#pragma unroll
      for (int l = 0; l < LIMBS; ++l) wv[u][l] = buffer_load_u4(wr, wvoff, soff + l * wlimb);
    }
  };
  auto store_w = [&]() {
    if (!w_thr) return;
#pragma unroll
    for (int u = 0; u < TPI; ++u)
#pragma unroll
      for (int l = 0; l < LIMBS; ++l) *reinterpret_cast<U4*>(sW + (u * LIMBS + l) * W_BYTES + wlds) = wv[u][l];
  };

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int kh = lane >> 5, l31 = lane & 31;
  // lane's output pixel of sub-tile j = (row wpix * NJ + j, column l31): plane row of its (ky = 0, kx = 0) input pixel
  int rbase[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) rbase[j] = S * (wpix * NJ + j) * S2_PLW + l31;
  const int abase = ((wco * MI) * 32 + l31) * S2_RB + (((kh ^ (l31 >> 3)) & 1) << 4);

  auto rescale_acc = [&](float f) {
    if (f != 1.f) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] *= f;
    }
  };
  // one tap (ky = iv, kx = u) of the chunk in LDS
  auto tap = [&](int iv, int u) {
    const int plane = S == 2 ? (u & 1) : 0, dcol = S == 2 ? (u >> 1) : u;       // kx = u
    bf16x8 fa[LIMBS][MI], fb[LIMBS][NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int row = rbase[j] + iv * S2_PLW + dcol;
      const int off = (row << 5) | ((((row >> 3) ^ plane ^ kh) & 1) << 4);
#pragma unroll
      for (int l = 0; l < LIMBS; ++l)
        fb[l][j] = *reinterpret_cast<const bf16x8*>(sP + l * LIMB_BYTES + plane * PLANE_BYTES + off);
    }
#pragma unroll
    for (int l = 0; l < LIMBS; ++l)
#pragma unroll
      for (int i = 0; i < MI; ++i)
        fa[l][i] = *reinterpret_cast<const bf16x8*>(sW + (u * LIMBS + l) * W_BYTES + abase + i * 32 * S2_RB);
    // smallest terms first
#pragma unroll
    for (int sum = LIMBS - 1; sum >= 0; --sum)
#pragma unroll
      for (int la = 0; la <= sum; ++la) {
        const int lb = sum - la;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = L::mfma(fa[la][i], fb[lb][j], acc[i][j]);
      }
  };

  // (Measured on the compiler, not on the GPU: finishing the next chunk's registers in front of the chunk's last barrier
  // interval and splitting them next to that interval's MFMAs - so that the top of a chunk only writes LDS - keeps the
  // 40 packed words alive across the loop's back edge beside the 40 in-flight loads: 56 - 163 spilled VGPRs in every
  // instantiation.  Not kept.)
  if (chunk0 < chunk1) {
    load_patch(chunk0);
    load_w(chunk0, 0);
    if (F16) prep_patch(chunk0);          // published by the chunk loop's first barrier
    for (int chunk = chunk0; chunk < chunk1; ++chunk) {
      const bool more = chunk + 1 < chunk1;
      __syncthreads();                    // the previous chunk's readers are done with sP
      if (F16) rescale_acc(block_exp_update(bexp, read_block_amax<NT / 64>(sAmax), a.exp_lo));
      else prep_patch(chunk);
      split_patch();
      write_patch();
      if (more) load_patch(chunk + 1);
#pragma unroll
      for (int iv = 0; iv < 3; ++iv) {
        store_w();
        __syncthreads();
        if (iv + 1 < 3) load_w(chunk, iv + 1);
        else if (more) load_w(chunk + 1, 0);
#pragma unroll
        for (int u = 0; u < TPI; ++u) tap(iv, u);
        __syncthreads();                  // sW may be overwritten (and, after the last interval, sP)
      }
      // binary16 limbs: the next chunk's registers landed during the nine taps; published by the loop's top barrier
      if (F16 && more) prep_patch(chunk + 1);
    }
  }
  const float esc = F16 ? exp2i(bexp.e) : 1.f;          // undo the block exponent (exact)

  const int ochan0 = (pn * a.groups + g) * a.cout_g;
  const float* osc = a.out_scale ? a.out_scale + ochan0 : nullptr;
  const float* bia = a.bias ? a.bias + g * a.cout_g : nullptr;
  if (a.part) {               // split-K: raw partial sums; splitk_reduce_kernel finishes (scale, bias, activation)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int oy = y0 + wpix * NJ + j, ox = x0 + l31;
      float* yp = a.part + (size_t)split * a.part_stride + (size_t)ochan0 * ohw + (size_t)oy * a.ow + ox;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + (wco * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (co >= a.cout_g) continue;
          yp[(size_t)co * ohw] = acc[i][j][r] * esc;
        }
      }
    }
    return;
  }
  // Epilogue as conv3x3_patch_kernel's: each wave transposes its 32co x 64pix sub-tile (two output rows of 32 pixels)
  // through LDS and stores 16 B per lane; everything it reads from global memory goes to LDS before the first store
  // (VMEM loads and stores share the in-order vmcnt), and LDS reads are grouped in front of the staging writes.
  __syncthreads();                                          // sP / sW are dead from here on
  float* stage = reinterpret_cast<float*>(smem) + wid * (32 * 64);
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
    const int oy = y0 + (p >> 5), ox = x0 + (p & (S2_TW - 1));
    ep_noise[p] = (a.act && a.act_noise) ? a.act_noise[(size_t)pn * ohw + (size_t)oy * a.ow + ox] : 0.f;
  }
  const float anw = (a.act && a.act_noise) ? a.act_noise_w[0] : 0.f;
  __syncthreads();
  const int pq = wpix * 64 + (lane & 15) * 4;                 // this lane's four pixels in every store of the loop below
  const float4 nz = *reinterpret_cast<const float4*>(ep_noise + pq);
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
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = it * 64 + lane;
      const int row = idx >> 4, c4 = idx & 15;
      const int co = co0 + (wco * MI + i) * 32 + row;
      const int p = wpix * 64 + c4 * 4;
      const int oy = y0 + (p >> 5), ox = x0 + (p & (S2_TW - 1));
      if (co < a.cout_g) {
        float4 v4 = v4s[it];
        if (a.act) {             // (NoiseInjection +) bias + leaky ReLU (fused_act.py:74-97)
          const float ab = abv[it];
          float t;
          t = v4.x + anw * nz.x + ab; v4.x = (t > 0.f ? t : t * a.act_alpha) * a.act_gain;
          t = v4.y + anw * nz.y + ab; v4.y = (t > 0.f ? t : t * a.act_alpha) * a.act_gain;
          t = v4.z + anw * nz.z + ab; v4.z = (t > 0.f ? t : t * a.act_alpha) * a.act_gain;
          t = v4.w + anw * nz.w + ab; v4.w = (t > 0.f ? t : t * a.act_alpha) * a.act_gain;
        }
        float* dst = a.y + (size_t)(ochan0 + co) * ohw + (size_t)oy * a.ow + ox;
        if (a.nt_store) __builtin_nontemporal_store(f32x4{v4.x, v4.y, v4.z, v4.w}, reinterpret_cast<f32x4*>(dst));
        else *reinterpret_cast<float4*>(dst) = v4;
      }
    }
    wave_lds_sync();                // the staging rows are this wave's own
  }
}

}  // namespace

namespace gg_conv {

bool s2_patch_serves(const ConvArgs& a, int tpix) {
  const int th = tpix / S2_TW;
  return a.ow >= S2_TW && a.ow % S2_TW == 0 && a.oh >= th && a.oh % th == 0 && a.cin_g % S2_CH == 0 &&
         a.cin_g <= kS2MaxCin && (long long)a.cin_g * a.h * a.w * 4 < (1LL << 31) &&
         (reinterpret_cast<uintptr_t>(a.y) & 15) == 0 && (!a.in_scale || (reinterpret_cast<uintptr_t>(a.in_scale) & 3) == 0);
}

#define S2_LAUNCH(ST, SC, TP, F) conv3x3s2_patch_kernel<ST, SC, TP, F><<<grid, TP * 2, 0, st>>>(a)
#define S2_LAUNCH_F(ST, SC, TP) do { if (a.f16) S2_LAUNCH(ST, SC, TP, true); else S2_LAUNCH(ST, SC, TP, false); } while (0)
void s2_patch_launch(const ConvArgs& a, int stride, int tpix, dim3 grid, hipStream_t st) {
  const bool sc = a.in_scale != nullptr;
  if (stride == 1) {                     // (128-pixel tiles only)
    if (sc) S2_LAUNCH_F(1, true, 128);
    else S2_LAUNCH_F(1, false, 128);
  } else if (tpix == 256) {
    if (sc) S2_LAUNCH_F(2, true, 256);
    else S2_LAUNCH_F(2, false, 256);
  } else {
    if (sc) S2_LAUNCH_F(2, true, 128);
    else S2_LAUNCH_F(2, false, 128);
  }
}

}  // namespace gg_conv
