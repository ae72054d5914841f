// a3  3x3 / stride 2 / pad 0 weight gradient, row-streaming formulation (round 4).
//
//   dW[co, ci, ky, kx] = sum_{n, y, c} dy[n, co, y, c] * x[n, ci, 2y + ky, 2c + kx]
// (the STN's down-sampling ConvLayers, networks.py:455-480: Blur, then a stride-2 convolution without padding).
// conv_wgrad_split_kernel serves it by gathering - and converting - every x element once per tap column of the GEMM:
// 74 - 136 TF/s on the C2 shapes (profiles/r04_f_conv_layers.txt).  This is conv3x3_wgrad_rows_kernel's scheme for
// stride 2: a block owns a (TCO co) x (TCI ci) x 9-tap tile and walks DOWN a strip of 32 dy columns, one dy row per slab:
//   * the dy row segment (TCO x 32 px) is staged as [co][px] (k = px contiguous -> MFMA A operand);
//   * x lives in LDS as a rolling window of 3 input rows per ci, each row split into its COLUMN-PARITY planes
//     E[i] = x[2 c0 + 2i] (33 entries) and O[i] = x[2 c0 + 2i + 1] (32): a slab converts the TWO new rows 2y+1, 2y+2
//     (row 2y is the previous slab's row 2(y-1)+2), and every x element then feeds all the taps that touch it;
//   * the B fragment of tap (ky, kx) is 8 consecutive dy columns px .. px+7 of window row 2y + ky = x columns
//     2 px + kx .. step 2: kx = 0 is an aligned 16-byte read of E, kx = 1 of O, kx = 2 is E shifted by one entry
//     (v_alignbit with the next dword).
// Each wave owns 32 co x 32 ci and keeps nine 32x32 accumulators (one per tap): 54 MFMAs per slab and wave for
// 16 gathered x elements + 16 dy elements per lane.  K-splits (image, strip, row block) go to the workspace
// [tile][split][tap][co][ci] and are summed in split order by wgrad_reduce_kernel (conv_mfma.hip).
#include "conv_common.h"

namespace {

using namespace gg_conv;

constexpr int SX_SLOT = 144;          // bytes per window row: E plane 40 entries (33 used) + O plane 32 entries, bf16
constexpr int SX_O = 80;              // byte offset of the O plane inside a row
constexpr int SX_CI = 3 * SX_SLOT + 32;   // bytes per ci: 464 = 29 x 16 (odd -> conflict-free ds_read_b128 lane stride)

template <int LIMBS, int TCO, int TCI>
__global__ __launch_bounds__(256, 2) void conv3x3s2_wgrad_rows_kernel(const WgradArgs a, int segs, int rblocks,
                                                                      int rows_per_block, int units_per_block,
                                                                      float* __restrict__ ws) {
  static_assert(TCO * TCI == 4096, "four waves of 32 co x 32 ci");
  constexpr int WCI = TCI / 32;                       // ci waves; co waves = 4 / WCI
  constexpr int DPARTS = 256 / TCO, DPX = 32 / DPARTS;        // dy: threads per row, pixels per thread
  constexpr int XPT = 16 * TCI / 256;                 // x items (ci, new row, 8-column group) per thread
  __shared__ __attribute__((aligned(16))) unsigned char sD[LIMBS][TCO * ROWB];
  __shared__ __attribute__((aligned(16))) unsigned char sX[LIMBS][TCI * SX_CI];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wci = wid % WCI, wco = wid / WCI;
  const int tiles_ci = a.tiles_j;
  const unsigned ntiles = (unsigned)a.tiles_co * tiles_ci;
  const unsigned logical = gg::xcd_remap(blockIdx.x, ntiles);
  const int tile_co = logical % a.tiles_co, tile_ci = logical / a.tiles_co;
  const int co0 = tile_co * TCO, ci0 = tile_ci * TCI;
  const int g = blockIdx.z;
  const int hw = a.h * a.w, ohw = a.oh * a.ow;
  const int total_units = a.batch * segs * rblocks;
  int c0 = 0, y0 = 0, y1 = 0;
  const float* dsrc = nullptr;
  __amdgpu_buffer_rsrc_t xr = uniform_rsrc(a.x, 0);

  // ---- dy mover: row drow, pixels dpart*DPX .. +DPX
  const int drow = tid / DPARTS, dpart = tid % DPARTS;
  const bool d_ok = (co0 + drow) < a.cout_g;
  int dcols = 0;                                    // valid float4 groups of this thread (ragged last strip)
  auto set_unit = [&](int unit) {
    int sp = unit;
    const int rb = sp % rblocks; sp /= rblocks;
    const int sg = sp % segs;
    const int n = sp / segs;
    c0 = sg * 32;
    y0 = rb * rows_per_block;
    y1 = y0 + rows_per_block;
    if (y1 > a.oh) y1 = a.oh;
    // x through a buffer resource over the image-group's input: rows beyond the image use an out-of-range offset (-> 0);
    // columns beyond the row only meet dy columns that are staged as zeros
    xr = uniform_rsrc(a.x + ((size_t)(n * a.groups + g) * a.cin_g) * hw, a.cin_g * hw * 4);
    const size_t doff = ((size_t)(n * a.groups + g) * a.cout_g + (d_ok ? co0 + drow : 0)) * ohw + c0 + dpart * DPX;
    dsrc = a.dy + doff;
    const int left = a.ow - (c0 + dpart * DPX);       // dy columns from this thread's first one to the row's end
    dcols = left <= 0 ? 0 : (left >= DPX ? DPX / 4 : left / 4);
  };
  // ---- x mover: items (ci, new row r of the slab's two, group of 8 input columns); + one right-halo column (E[32])
  int xci[XPT], xr01[XPT], xg[XPT];
  bool x_ok[XPT];
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    const int item = tid + 256 * i;
    xg[i] = item & 7;
    xr01[i] = (item >> 3) & 1;
    xci[i] = item >> 4;
    x_ok[i] = (ci0 + xci[i]) < a.cin_g;
  }
  const bool h_thr = tid < 2 * TCI;                   // halo movers: (ci = tid >> 1, row = tid & 1)
  const int hci = tid >> 1, hr01 = tid & 1;
  const bool h_ok = h_thr && (ci0 + hci) < a.cin_g;

  float4 rd[DPX / 4];
  U4 rx[XPT][2];
  float rh = 0.f;
  auto load_dy = [&](int y) {
    const bool ok = d_ok & (y < y1);
#pragma unroll
    for (int q = 0; q < DPX / 4; ++q)
      rd[q] = (ok && q < dcols) ? *reinterpret_cast<const float4*>(dsrc + (size_t)y * a.ow + q * 4)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto load_x = [&](int rowbase) {                  // input rows rowbase, rowbase + 1
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int row = rowbase + xr01[i];
      const bool ok = ((unsigned)row < (unsigned)a.h) & x_ok[i];
      const unsigned vo = ok ? (unsigned)((ci0 + xci[i]) * hw + row * a.w + 2 * c0 + 8 * xg[i]) * 4u : kOobOffset;
      rx[i][0] = buffer_load_u4(xr, vo, 0);
      rx[i][1] = buffer_load_u4(xr, vo, 16);
    }
    if (h_thr) {
      const int row = rowbase + hr01;
      const bool ok = ((unsigned)row < (unsigned)a.h) & h_ok;
      rh = buffer_load_f32(xr, ok ? (unsigned)((ci0 + hci) * hw + row * a.w + 2 * c0 + 64) * 4u : kOobOffset, 0);
    }
  };
  auto store_dy = [&]() {
    float v[DPX];
#pragma unroll
    for (int q = 0; q < DPX / 4; ++q) { v[4 * q] = rd[q].x; v[4 * q + 1] = rd[q].y; v[4 * q + 2] = rd[q].z; v[4 * q + 3] = rd[q].w; }
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
      const int slot = (rowbase + xr01[i] + 3) % 3;
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // (through scalars: __builtin_bit_cast applied to a vector ELEMENT reads element 0 for every e with this compiler)
        const unsigned u0 = rx[i][0][e], u1 = rx[i][1][e];
        v[e] = __builtin_bit_cast(float, u0);
        v[4 + e] = __builtin_bit_cast(float, u1);
      }
#pragma unroll
      for (int l = 0; l < LIMBS; ++l) {
        unsigned char* base = &sX[l][xci[i] * SX_CI + slot * SX_SLOT];
        const unsigned e0 = pack_bf16x2(v[0], v[2]), e1 = pack_bf16x2(v[4], v[6]);      // even columns
        const unsigned o0 = pack_bf16x2(v[1], v[3]), o1 = pack_bf16x2(v[5], v[7]);      // odd columns
        *reinterpret_cast<uint2*>(base + xg[i] * 8) = make_uint2(e0, e1);
        *reinterpret_cast<uint2*>(base + SX_O + xg[i] * 8) = make_uint2(o0, o1);
        if (l + 1 < LIMBS) {
          v[0] -= bf16_lo(e0); v[2] -= bf16_hi(e0); v[4] -= bf16_lo(e1); v[6] -= bf16_hi(e1);
          v[1] -= bf16_lo(o0); v[3] -= bf16_hi(o0); v[5] -= bf16_lo(o1); v[7] -= bf16_hi(o1);
        }
      }
    }
    if (h_thr) {
      const int slot = (rowbase + hr01 + 3) % 3;
      float hv = rh;
#pragma unroll
      for (int l = 0; l < LIMBS; ++l) {
        const __bf16 hb = (__bf16)hv;
        *reinterpret_cast<__bf16*>(&sX[l][hci * SX_CI + slot * SX_SLOT + 32 * 2]) = hb;      // E[32]
        if (l + 1 < LIMBS) hv -= (float)hb;
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
  const int b_off = (wci * 32 + l31) * SX_CI + kh * 16;              // + slot*SX_SLOT + ks*32 (+ SX_O)

  for (int u = 0; u < units_per_block; ++u) {
    const int unit = blockIdx.y * units_per_block + u;
    if (unit >= total_units) break;
    set_unit(unit);
    if (y0 >= y1) continue;
    // prologue: input row 2 y0 (the pair starts one row early; that row's slot is rewritten by the first slab)
    load_x(2 * y0 - 1);
    store_x(2 * y0 - 1);
    load_dy(y0);
    load_x(2 * y0 + 1);
    for (int y = y0; y < y1; ++y) {
      store_dy();
      store_x(2 * y + 1);
      __syncthreads();
      load_dy(y + 1);
      load_x(2 * y + 3);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 fa[LIMBS];
#pragma unroll
        for (int l = 0; l < LIMBS; ++l) fa[l] = *reinterpret_cast<const bf16x8*>(&sD[l][a_off + ks * 32]);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int slot = (2 * y + ky) % 3;
          bf16x8 fb[LIMBS][3];
#pragma unroll
          for (int l = 0; l < LIMBS; ++l) {
            const unsigned char* p = &sX[l][b_off + slot * SX_SLOT + ks * 32];
            const U4 ev = *reinterpret_cast<const U4*>(p);
            const U4 od = *reinterpret_cast<const U4*>(p + SX_O);
            const unsigned next = *reinterpret_cast<const unsigned*>(p + 16);
            const U4 right{__builtin_amdgcn_alignbit(ev[1], ev[0], 16), __builtin_amdgcn_alignbit(ev[2], ev[1], 16),
                           __builtin_amdgcn_alignbit(ev[3], ev[2], 16), __builtin_amdgcn_alignbit(next, ev[3], 16)};
            fb[l][0] = __builtin_bit_cast(bf16x8, ev);                // kx = 0: columns 2 px
            fb[l][1] = __builtin_bit_cast(bf16x8, od);                // kx = 1: 2 px + 1
            fb[l][2] = __builtin_bit_cast(bf16x8, right);             // kx = 2: 2 px + 2
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

  // partial tile -> workspace [tile][split][tap][co][ci] (ci fastest: 128-byte runs per store); summed by
  // wgrad_reduce_kernel
  const size_t tile = ((size_t)g * a.tiles_co + tile_co) * tiles_ci + tile_ci;
  float* base = ws + (tile * gridDim.y + blockIdx.y) * (size_t)(9 * TCO * TCI);
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int col = wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      base[((size_t)t * TCO + col) * TCI + wci * 32 + l31] = acc[t][r];
    }
}

}  // namespace

namespace gg_conv {

void s2_wgrad_rows_launch(const WgradArgs& a, int limbs, bool narrow, int segs, int rblocks, int rows_per_block,
                          int units_per_block, float* ws, dim3 grid, hipStream_t st) {
  if (limbs == 1) {
    if (narrow) conv3x3s2_wgrad_rows_kernel<1, 64, 64><<<grid, 256, 0, st>>>(a, segs, rblocks, rows_per_block, units_per_block, ws);
    else conv3x3s2_wgrad_rows_kernel<1, 128, 32><<<grid, 256, 0, st>>>(a, segs, rblocks, rows_per_block, units_per_block, ws);
  } else if (limbs == 2) {
    if (narrow) conv3x3s2_wgrad_rows_kernel<2, 64, 64><<<grid, 256, 0, st>>>(a, segs, rblocks, rows_per_block, units_per_block, ws);
    else conv3x3s2_wgrad_rows_kernel<2, 128, 32><<<grid, 256, 0, st>>>(a, segs, rblocks, rows_per_block, units_per_block, ws);
  } else {
    if (narrow) conv3x3s2_wgrad_rows_kernel<3, 64, 64><<<grid, 256, 0, st>>>(a, segs, rblocks, rows_per_block, units_per_block, ws);
    else conv3x3s2_wgrad_rows_kernel<3, 128, 32><<<grid, 256, 0, st>>>(a, segs, rblocks, rows_per_block, units_per_block, ws);
  }
}

}  // namespace gg_conv
