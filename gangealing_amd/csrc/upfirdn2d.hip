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
#include <cstdlib>

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
  // round 6: the activation's sign as ONE BIT per element in the stream kernel's own tiling of the FORWARD output
  // (strips of SW_OUT = 61 columns x chunks of SROWS = 16 rows): uint64 [plane][chunk_y][strip_x][row in chunk], bit j =
  // column strip_x * 61 + j.  A wave of the forward owns one (chunk, strip) cell and writes its 16 words with one
  // 128-byte store (no atomics, nothing to clear); a lane of the adjoint (PRO = 2) reads the 32-bit half that holds its
  // column - a 64-lane row load touches <= 24 bytes instead of 256.  bit_strips / bit_chunks: the forward's tiling.
  unsigned long long* sign_bits = nullptr;
  const unsigned* ref_bits = nullptr;
  int bit_strips = 0, bit_chunks = 0;
  // EPI = 2 (round 6): the blur is the ADJOINT of a Blur that followed a conv + leaky-ReLU layer (ResBlock's conv1 -> Blur,
  // networks.py:375-386) and the activation's backward rides in its epilogue: out = lrelu'(out_ref) * blur(in), out_ref =
  // the layer's saved output (the shape of `out`); wave_sums[unit] receives the sum of the strip's outputs - the bias
  // gradient's partial sums, added per channel in a fixed order by blur_bias_reduce_kernel.
  const float* out_ref = nullptr;
  float* wave_sums = nullptr;
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

// ---------------------------------------------------------------------------------------------------------------
// Streaming blur (up = down = 1, 4x4 taps), the default for the hot case.  No LDS, no barrier, no integer division:
//   * one WAVE owns a strip of 61 output columns x SROWS output rows of one plane.  Lane l holds input column
//     x0 + l (64 columns = 61 outputs + the 3-column halo); a row is one coalesced 256-byte wave load.
//   * the horizontal 4-tap pass happens in registers when a row arrives: the three neighbours to the right come from
//     `v_mov_b32_dpp ... wave_shl:1` (whole-wave shift, lane i <- lane i+1, zero shifted into lane 63), applied
//     1x / 2x / 3x.  The vertical pass is a 4-deep register window sliding down the strip, so every input word is
//     loaded exactly once per strip and every output row leaves as one 244-byte coalesced store.
//   * rows are fetched eight at a time (eight independent loads in flight per wave) before they are consumed.
// Separable taps (every Blur of the networks: make_kernel is an outer product, networks.py:34-41) cost 3 DPP moves +
// 8 FMAs per output; general taps keep the three shifted copies of each window row (16 FMAs per output).
// Read amplification: 64/61 columns x (SROWS+3)/SROWS rows = 1.25, served by L2 (neighbouring strips run together).
// Strip height, measured on the three hot shapes at batch 16 (TB/s of in + out bytes): 8 rows 4.0-4.7 / 3.8 / 4.6,
// 16 rows 4.67 / 4.70 / 4.49, 32 rows 4.45 / 4.35 / 4.26, 64 rows 4.19 / 4.17 / 3.57, 128 rows 3.94 / 3.90 / 3.21:
// taller strips save halo reads that L2 already absorbs and lose memory-level parallelism (fewer waves per plane).
constexpr int SW_OUT = 61;      // outputs per wave row
constexpr int SROWS = 16;       // output rows per wave

__device__ __forceinline__ float wave_shl1(float v) {     // lane i <- lane i+1; lane 63 <- 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// PRO: 0 = plain input; 1 = input * lrelu'(ref) from the saved fp32 output; 2 = the same factor from the sign plane
// EPI: 0 = plain store; 1 = + noise + bias, leaky ReLU (the up-sampling StyledConv's tail); 2 = * lrelu'(out_ref) (+ strip sums)
template <int EPI, int PRO>
__global__ __launch_bounds__(256) void upfirdn2d_blur4_stream(
    float* __restrict__ out, const float* __restrict__ in, const float* __restrict__ kernel,
    int in_h, int in_w, int out_h, int out_w, int pad_x0, int pad_y0,
    int strips_x, int chunks_y, unsigned nunits, const BlurFuse f) {
  const int lane = threadIdx.x & 63;
  // the wave index goes through readfirstlane: everything derived from it (plane, rows, buffer descriptors, scalar
  // offsets) is then provably wave-uniform and lives in SGPRs
  const unsigned unit = blockIdx.x * 4u + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (unit >= nunits) return;                               // whole wave leaves (no barriers in this kernel)
  const int sx = (int)(unit % (unsigned)strips_x);
  const unsigned q = unit / (unsigned)strips_x;
  const int cy = (int)(q % (unsigned)chunks_y);
  const unsigned plane = q / (unsigned)chunks_y;
  float kf[16];                                             // flipped taps (wave-uniform scalar loads)
#pragma unroll
  for (int i = 0; i < 16; ++i) kf[i] = kernel[15 - i];
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

  const int ox = sx * SW_OUT + lane;
  const int ix = ox - pad_x0;                               // input column held by this lane
  const int oy0 = cy * SROWS;
  const int iy0 = oy0 - pad_y0;                             // input row of tap 0 for output row oy0
  // bounds live in the buffer offsets (out-of-range -> load 0 / store dropped), not in branches: the eight row loads
  // of a batch issue back to back
  const unsigned in_off = (ix >= 0 && ix < in_w) ? (unsigned)ix * 4u : gg::kOobOffset;
  const unsigned out_off = (lane < SW_OUT && ox < out_w) ? (unsigned)ox * 4u : gg::kOobOffset;
  const int in_bytes = in_h * in_w * 4, out_bytes = out_h * out_w * 4;
  const __amdgpu_buffer_rsrc_t src = gg::uniform_rsrc(in + (size_t)plane * in_h * in_w, in_bytes);
  const __amdgpu_buffer_rsrc_t rsrc = gg::uniform_rsrc(PRO == 1 ? f.ref + (size_t)plane * in_h * in_w : in, in_bytes);
  // PRO == 2: the 32-bit half of the forward's (chunk, strip) word that holds this lane's column (neighbouring lanes share it)
  const int bits_bytes = f.bit_chunks * f.bit_strips * SROWS * 8;
  const __amdgpu_buffer_rsrc_t brsrc = gg::uniform_rsrc(
      PRO == 2 ? reinterpret_cast<const float*>(f.ref_bits + (size_t)plane * (bits_bytes / 4)) : in,
      PRO == 2 ? bits_bytes : in_bytes);
  const int bstrip = ix >= 0 ? ix / SW_OUT : 0, bcol = ix - bstrip * SW_OUT;
  const unsigned bit_off = (ix >= 0 && ix < in_w) ? (unsigned)(bstrip * SROWS * 8 + (bcol >> 5) * 4) : gg::kOobOffset;
  const int bit_sh = bcol & 31;
  const __amdgpu_buffer_rsrc_t dst = gg::uniform_rsrc(out + (size_t)plane * out_h * out_w, out_bytes);
  float nw = 0.f, ab = 0.f;
  const int n_img = EPI == 1 ? (int)(plane / f.channels) : 0;
  // the epilogue's second operand at the OUTPUT position: the noise image (EPI = 1) or the saved activation (EPI = 2)
  const __amdgpu_buffer_rsrc_t nz = gg::uniform_rsrc(
      EPI == 1 ? f.noise + (size_t)n_img * out_h * out_w : (EPI == 2 ? f.out_ref + (size_t)plane * out_h * out_w : in),
      out_bytes);
  float wsum = 0.f;                                         // EPI = 2: sum of this lane's outputs
  if (EPI == 1) {
    nw = f.noise_w[0];
    ab = f.bias[plane - (unsigned)n_img * (unsigned)f.channels];
  }
  auto fetch = [&](int iy) -> float {
    const bool row_ok = iy >= 0 && iy < in_h;               // wave-uniform
    const unsigned vo = row_ok ? in_off : gg::kOobOffset;
    const int so = row_ok ? iy * in_w * 4 : 0;
    float v = gg::buffer_load_f32(src, vo, so);
    if (PRO == 1) v *= (gg::buffer_load_f32(rsrc, vo, so) > 0.f) ? f.gain : f.gain * f.alpha;
    if (PRO == 2) {
      // (iy >> 4, iy & 15: chunk and row of the forward's tiling; wave-uniform -> the scalar offset)
      const unsigned wd = __builtin_bit_cast(
          unsigned, gg::buffer_load_f32(brsrc, row_ok ? bit_off : gg::kOobOffset,
                                        row_ok ? ((iy >> 4) * f.bit_strips * SROWS + (iy & 15)) * 8 : 0));
      v *= ((wd >> bit_sh) & 1u) ? f.gain : f.gain * f.alpha;
    }
    return v;
  };
  // noise of output row oy at this lane's column (EPI); loaded with the rows, BEFORE the strip's first store: VMEM loads
  // and stores share the in-order vmcnt, so a load issued after a store is waited for together with the store's
  // acknowledgement (the round-2 form loaded the noise inside `finish`: one store round trip per output row)
  auto noise_at = [&](int oy) -> float {
    const bool row_ok = oy < out_h;
    return gg::buffer_load_f32(nz, row_ok ? out_off : gg::kOobOffset, row_ok ? oy * out_w * 4 : 0);
  };
  unsigned long long my_bits = 0ull;                       // EPI + sign plane: lane r keeps row r's word of this cell
  auto finish = [&](float acc, int oy, float noise, int u) {
    const bool row_ok = oy < out_h;
    const unsigned vo = row_ok ? out_off : gg::kOobOffset;
    const int so = row_ok ? oy * out_w * 4 : 0;
    if (EPI == 1) {
      const float t = acc + nw * noise + ab;
      acc = (t > 0.f ? t : t * f.alpha) * f.gain;
    }
    if (EPI == 2) {                                         // fused_act.py:33-38 on the blurred gradient (`noise` = out_ref here)
      acc = ((noise > 0.f) ? acc : acc * f.alpha) * f.gain;
      if (row_ok && lane < SW_OUT && ox < out_w) wsum += acc;
    }
    gg::buffer_store_f32(acc, dst, vo, so);
    if (EPI == 1 && f.sign_bits) {                          // (wave-uniform) exactly the test the backward applies to `out`
      const unsigned long long m = __ballot(row_ok && lane < SW_OUT && ox < out_w && acc > 0.f);
      if (lane == u) my_bits = m;
    }
  };
  auto flush_bits = [&]() {
    if (EPI == 1 && f.sign_bits && lane < SROWS)
      f.sign_bits[(((size_t)plane * f.bit_chunks + cy) * f.bit_strips + sx) * SROWS + lane] = my_bits;
    if (EPI == 2 && f.wave_sums) {                          // fixed shuffle tree: the same sum on every run
      const float tot = gg::wave_sum(wsum);
      if (lane == 0) f.wave_sums[unit] = tot;
    }
  };
  if (sep) {
    auto hpass = [&](float v) -> float {
      const float s1 = wave_shl1(v), s2 = wave_shl1(s1), s3 = wave_shl1(s2);
      // explicit FMA chains (here and in the vertical pass): every instantiation of this kernel rounds the same way,
      // whatever the compiler's contraction choices around the masked / plain loads (route-equivalence tests are bitwise)
      return fmaf(s3, kb[3], fmaf(s2, kb[2], fmaf(s1, kb[1], v * kb[0])));
    };
    // every load of the strip (3 + SROWS input rows, SROWS noise rows) is issued before its first store
    const float r0 = fetch(iy0), r1 = fetch(iy0 + 1), r2 = fetch(iy0 + 2);
    float nv[SROWS], nzv[SROWS];
#pragma unroll
    for (int u = 0; u < SROWS; ++u) nv[u] = fetch(iy0 + 3 + u);
#pragma unroll
    for (int u = 0; u < SROWS; ++u) nzv[u] = EPI != 0 ? noise_at(oy0 + u) : 0.f;
    float h0 = hpass(r0), h1 = hpass(r1), h2 = hpass(r2);
#pragma unroll
    for (int u = 0; u < SROWS; ++u) {
      const float h3 = hpass(nv[u]);
      finish(fmaf(h3, ka[3], fmaf(h2, ka[2], fmaf(h1, ka[1], h0 * ka[0]))), oy0 + u, nzv[u], u);
      h0 = h1; h1 = h2; h2 = h3;
    }
    flush_bits();
    return;
  }
  // general taps: window[row][shift]
  float w[4][4];
  auto shifts = [&](float v, float s[4]) {
    s[0] = v; s[1] = wave_shl1(v); s[2] = wave_shl1(s[1]); s[3] = wave_shl1(s[2]);
  };
  shifts(fetch(iy0), w[0]);
  shifts(fetch(iy0 + 1), w[1]);
  shifts(fetch(iy0 + 2), w[2]);
  for (int rr = 0; rr < SROWS; ++rr) {
    if (oy0 + rr >= out_h) break;
    shifts(fetch(iy0 + 3 + rr), w[3]);
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc += w[r][c] * kf[r * 4 + c];
    finish(acc, oy0 + rr, EPI != 0 ? noise_at(oy0 + rr) : 0.f, rr);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) w[r][c] = w[r + 1][c];
  }
  flush_bits();
}

// A/B switch for measurements: GG_BLUR_TILE=1 selects the round-1 LDS tile kernel.
inline bool blur_use_tile_kernel() {
  static const bool v = [] { const char* e = getenv("GG_BLUR_TILE"); return e && e[0] == '1'; }();
  return v;
}

template <int EPI, int PRO>
int launch_blur4(float* out, const float* in, const float* kernel, long long planes, int in_h, int in_w, int out_h,
                 int out_w, int pad_x0, int pad_y0, const BlurFuse& f, hipStream_t st) {
  if (blur_use_tile_kernel() && PRO != 2 && EPI != 2 && !f.sign_bits) {
    const int tiles_x = (out_w + TILE - 1) / TILE, tiles_y = (out_h + TILE - 1) / TILE;
    const long long ntiles = (long long)tiles_x * tiles_y * planes;
    if (ntiles >= (1LL << 31)) return gg::fail(-2, "blur4: too many tiles");
    upfirdn2d_blur4_tile<EPI == 1, PRO == 1><<<(unsigned)ntiles, 256, 0, st>>>(out, in, kernel, (int)planes, in_h, in_w, out_h,
                                                                    out_w, pad_x0, pad_y0, tiles_x, tiles_y,
                                                                    (unsigned)ntiles, f);
    return gg::launch_status("upfirdn2d_blur4_tile");
  }
  const int strips_x = (out_w + SW_OUT - 1) / SW_OUT, chunks_y = (out_h + SROWS - 1) / SROWS;
  const long long nunits = (long long)strips_x * chunks_y * planes;
  if (nunits >= (1LL << 31)) return gg::fail(-2, "blur4: too many strips");
  const long long blocks = (nunits + 3) / 4;
  upfirdn2d_blur4_stream<EPI, PRO><<<(unsigned)blocks, 256, 0, st>>>(out, in, kernel, in_h, in_w, out_h, out_w, pad_x0,
                                                                     pad_y0, strips_x, chunks_y, (unsigned)nunits, f);
  return gg::launch_status("upfirdn2d_blur4_stream");
}

constexpr int MAX_LDS_TAPS = 1024;

// K = tap / accumulator type: T itself for float and double; float for binary16 tensors (the reference's kernel
// accumulates in scalar_t, upfirdn2d_kernel.cu:191-201; one rounding at the end is the tighter result)
// Thread -> one output position (oy, ox) of the plane, 32-bit index math (one unsigned division per thread; the round-2
// form decomposed a 64-bit flat index with two 64-bit divisions per output - several hundred VALU instructions for a
// 16-tap filter); the tap geometry depends on the position only, so it is computed once and the thread then walks over
// planes blockIdx.y, blockIdx.y + gridDim.y, ...
// FAST: 0 = any up / down / taps; 1 = 4x4 taps, up 1, down 2 (the ResBlock skip's blur + stride-2 read-out and the
// adjoint of the ToRGB up-sampling); 2 = 4x4 taps, up 2, down 1 (ToRGB skip up-sampling, adjoint of the former); 3 (round
// 6) = 4x4 taps, up = down = 1 on planes too small for the streaming blur (the generator's and the STN's blurs at <= 16^2:
// the generic loop's per-tap branches and 64-bit offsets ran them at 0.74 TB/s): the tap loops have constant trip counts there.
template <typename T, typename K = T, int FAST = 0>
__global__ __launch_bounds__(256) void upfirdn2d_direct(
    T* __restrict__ out, const T* __restrict__ in, const K* __restrict__ kernel,
    int major, int in_h, int in_w, int out_h, int out_w, int kh_, int kw_,
    int up_x_, int up_y_, int down_x_, int down_y_, int pad_x0, int pad_y0, const T* __restrict__ addend = nullptr) {
  const int kh = FAST ? 4 : kh_, kw = FAST ? 4 : kw_;
  const int up_x = (FAST == 1 || FAST == 3) ? 1 : FAST == 2 ? 2 : up_x_, up_y = (FAST == 1 || FAST == 3) ? 1 : FAST == 2 ? 2 : up_y_;
  const int down_x = FAST == 1 ? 2 : (FAST == 2 || FAST == 3) ? 1 : down_x_;
  const int down_y = FAST == 1 ? 2 : (FAST == 2 || FAST == 3) ? 1 : down_y_;
  __shared__ K sk[MAX_LDS_TAPS];
  const bool lds_taps = kh * kw <= MAX_LDS_TAPS;
  if (lds_taps) {
    for (int i = threadIdx.x; i < kh * kw; i += 256) {
      const int ky = i / kw, kx = i - ky * kw;
      sk[i] = kernel[(kh - 1 - ky) * kw + (kw - 1 - kx)];
    }
    __syncthreads();
  }
  const unsigned per_out = (unsigned)out_h * (unsigned)out_w;
  const size_t per_in = (size_t)in_h * in_w;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < per_out; idx += gridDim.x * 256u) {
    const int oy = (int)(idx / (unsigned)out_w);
    const int ox = (int)(idx - (unsigned)oy * (unsigned)out_w);
    const int mid_x = ox * down_x + up_x - 1 - pad_x0;
    const int mid_y = oy * down_y + up_y - 1 - pad_y0;
    const int ix0 = gg::floor_div(mid_x, up_x), iy0 = gg::floor_div(mid_y, up_y);
    const int tx0 = ix0 * up_x + pad_x0 - ox * down_x;
    const int ty0 = iy0 * up_y + pad_y0 - oy * down_y;
    if (FAST) {
      // constant trip counts: NT taps per axis starting at (tx0, ty0) in {0 .. up - 1}; a tap outside the image reads a
      // clamped (always valid) address and is replaced by zero - no branches in the plane loop
      constexpr int NT = FAST == 2 ? 2 : 4, UP = FAST == 2 ? 2 : 1;
      K kv[NT][NT];
      int off[NT][NT];
      unsigned okm = 0;
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) {
          const int iy = iy0 + a, ix = ix0 + b, ty = ty0 + a * UP, tx = tx0 + b * UP;
          const bool ok = (unsigned)iy < (unsigned)in_h && (unsigned)ix < (unsigned)in_w && ty < 4 && tx < 4;
          kv[a][b] = ok ? sk[ty * 4 + tx] : K(0);
          off[a][b] = ok ? iy * in_w + ix : 0;
          okm |= (ok ? 1u : 0u) << (a * NT + b);
        }
      for (int plane = blockIdx.y; plane < major; plane += gridDim.y) {
        const T* src = in + (size_t)plane * per_in;
        // same summation order as the generic loop (rows outer, columns inner), skipped taps add an exact zero
        K acc = K(0);
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
          for (int b = 0; b < NT; ++b) {
            const K v = K(src[off[a][b]]);
            acc += ((okm >> (a * NT + b)) & 1u ? v : K(0)) * kv[a][b];
          }
        const size_t o = (size_t)plane * per_out + idx;
        if (addend) acc += K(addend[o]);
        out[o] = T(acc);
      }
      continue;
    }
    for (int plane = blockIdx.y; plane < major; plane += gridDim.y) {
      const T* src = in + (size_t)plane * per_in;
      K acc = K(0);
      for (int ty = ty0, iy = iy0; ty < kh; ty += up_y, ++iy) {
        if (iy < 0 || iy >= in_h) continue;
        for (int tx = tx0, ix = ix0; tx < kw; tx += up_x, ++ix) {
          if (ix < 0 || ix >= in_w) continue;
          const K kv = lds_taps ? sk[ty * kw + tx] : kernel[(kh - 1 - ty) * kw + (kw - 1 - tx)];
          acc += K(src[(size_t)iy * in_w + ix]) * kv;
        }
      }
      const size_t o = (size_t)plane * per_out + idx;
      if (addend) acc += K(addend[o]);
      out[o] = T(acc);
    }
  }
}

// 4x4 FIR, up 1, down 2 as a row walk (the ResBlock skip's blur evaluated only where its 1x1 / stride-2 convolution
// reads, networks.py:455-480 + 508-515).  upfirdn2d_direct<.., 1> issues 16 four-byte loads per output (1.3 TB/s on
// 64 -> 64 channels @128^2); here a wave owns a strip of output rows, a lane one output column (narrow images: several
// planes side by side in a wave), and the 4 x 4 input window slides down two rows per output row: two 8-byte loads
// per new input row = four load instructions per output.  Rows / columns outside the image are exact zeros (buffer
// range check for the rows, a select for the columns); the sum runs in the order of the generic kernel (rows outer,
// columns inner), so the results are bit-identical to it.
constexpr int D2_ROWS = 8;                    // output rows per wave
__global__ __launch_bounds__(256) void upfirdn2d_fir4_down2_kernel(float* __restrict__ out, const float* __restrict__ in,
                                                                   const float* __restrict__ kernel, int major, int in_h,
                                                                   int in_w, int out_h, int out_w, int pad_x0, int pad_y0,
                                                                   int w2_log2, int strips, int cchunks,
                                                                   long long nunits) {
  const int lane = threadIdx.x & 63;
  const long long unit = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= nunits) return;                               // whole wave leaves (no barriers in this kernel)
  // unit -> (plane group, row strip, column chunk); lane -> (plane of the group, column)
  const int cc = (int)(unit % cchunks);
  const long long u2 = unit / cchunks;
  const int strip = (int)(u2 % strips);
  const int pg = (int)(u2 / strips);
  const int w2 = 1 << w2_log2;                              // columns per plane inside a wave (<= 64)
  const int ppw = 64 >> w2_log2;                            // planes per wave
  const int plane = pg * ppw + (lane >> w2_log2);
  const int ox = cc * 64 + (lane & (w2 - 1));
  const bool lane_ok = plane < major && ox < out_w;
  float kv[4][4];                                           // flipped taps (upfirdn2d_kernel.cu:149-206)
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) kv[a][b] = kernel[(3 - a) * 4 + (3 - b)];
  const int ix0 = 2 * ox - pad_x0;
  bool cok[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) cok[b] = lane_ok && (unsigned)(ix0 + b) < (unsigned)in_w;
  const __amdgpu_buffer_rsrc_t rs = gg::uniform_rsrc(in, (int)((long long)major * in_h * in_w * 4));
  // The row's four columns ix0 .. ix0 + 3 as 8-byte loads that start on EVEN columns (ix0's parity is that of the
  // padding: uniform).  A pair that starts left of the image is wholly outside it (columns -2, -1), so an offset that
  // wraps below zero - out of range as unsigned, reads 0 - only ever stands for zeros; everything else is selected by
  // the per-column validity.  Odd ix0: scalar, pair, scalar.
  const bool odd = (pad_x0 & 1) != 0;
  const long long pbase = (long long)plane * in_h * in_w + ix0;         // element index of (row 0, column ix0)
  const int y_begin = strip * D2_ROWS;
  int y_end = y_begin + D2_ROWS;
  if (y_end > out_h) y_end = out_h;
  float win[4][4];
  typedef float f2 __attribute__((ext_vector_type(2)));
  auto load_row = [&](float (&r)[4], int iy) {
    if ((unsigned)iy < (unsigned)in_h) {                    // uniform: rows are per wave
      const long long e = pbase + (long long)iy * in_w;
      float c0, c1, c2, c3;
      if (odd) {
        const unsigned v0 = cok[0] ? (unsigned)(e * 4) : gg::kOobOffset;
        const unsigned v1 = lane_ok ? (unsigned)((e + 1) * 4) : gg::kOobOffset;
        const unsigned v3 = cok[3] ? (unsigned)((e + 3) * 4) : gg::kOobOffset;
        c0 = gg::buffer_load_f32(rs, v0, 0);
        const f2 mid = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)v1, 0, 0));
        c3 = gg::buffer_load_f32(rs, v3, 0);
        c1 = mid[0]; c2 = mid[1];
      } else {
        // (each pair has its own lane offset: the range check sees the lane offset only, and with ix0 = -2 the first
        // pair lies in front of the row while the second one is columns 0, 1)
        const unsigned v0 = lane_ok ? (unsigned)(e * 4) : gg::kOobOffset;
        const unsigned v2 = lane_ok ? (unsigned)((e + 2) * 4) : gg::kOobOffset;
        const f2 lo = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)v0, 0, 0));
        const f2 hi = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)v2, 0, 0));
        c0 = lo[0]; c1 = lo[1]; c2 = hi[0]; c3 = hi[1];
      }
      r[0] = cok[0] ? c0 : 0.f; r[1] = cok[1] ? c1 : 0.f; r[2] = cok[2] ? c2 : 0.f; r[3] = cok[3] ? c3 : 0.f;
    } else {
      r[0] = r[1] = r[2] = r[3] = 0.f;
    }
  };
  const int iy_first = 2 * y_begin - pad_y0;
  load_row(win[0], iy_first);
  load_row(win[1], iy_first + 1);
  for (int oy = y_begin; oy < y_end; ++oy) {
    const int iy0 = 2 * oy - pad_y0;
    load_row(win[2], iy0 + 2);
    load_row(win[3], iy0 + 3);
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc += win[a][b] * kv[a][b];
    if (lane_ok) out[((size_t)plane * out_h + oy) * out_w + ox] = acc;
#pragma unroll
    for (int b = 0; b < 4; ++b) { win[0][b] = win[2][b]; win[1][b] = win[3][b]; }
  }
}

// true when the row-walk kernel took the launch
bool launch_fir4_down2(float* out, const float* in, const float* kernel, int major, int in_h, int in_w, int out_h,
                       int out_w, int pad_x0, int pad_y0, hipStream_t st) {
  static const bool off = getenv("GG_NO_FIR4_DOWN2") != nullptr;          // measurement switch
  if (off || (long long)major * in_h * in_w * 4 >= (1LL << 31) || out_w < 4) return false;
  int w2_log2 = 0;
  while ((1 << w2_log2) < out_w && w2_log2 < 6) ++w2_log2;
  const int ppw = 64 >> w2_log2;
  const int cchunks = (out_w + 63) / 64, strips = (out_h + D2_ROWS - 1) / D2_ROWS;
  const long long nunits = (long long)((major + ppw - 1) / ppw) * strips * cchunks;
  const long long blocks = (nunits + 3) / 4;
  if (blocks >= (1LL << 31)) return false;
  upfirdn2d_fir4_down2_kernel<<<(unsigned)blocks, 256, 0, st>>>(out, in, kernel, major, in_h, in_w, out_h, out_w, pad_x0,
                                                               pad_y0, w2_log2, strips, cchunks, nunits);
  return true;
}

// grid: x covers one plane's outputs, y strides over planes (<= 64 planes per thread)
// A/B switch for measurements: GG_NO_FIR4_SMALL=1 keeps the generic loop for small-plane 4x4 blurs.
inline bool fir4_small_fast() {
  static const bool v = getenv("GG_NO_FIR4_SMALL") == nullptr;
  return v;
}

template <typename T, typename K>
int launch_direct(T* out, const T* in, const K* kernel, const T* addend, int major, int in_h, int in_w, int out_h,
                  int out_w, int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_y0,
                  hipStream_t st, const char* what) {
  if ((long long)out_h * out_w >= (1LL << 31) || (long long)in_h * in_w >= (1LL << 31))
    return gg::fail(-2, "upfirdn2d: a single plane has 2^31 or more elements");
  const unsigned per_out = (unsigned)out_h * (unsigned)out_w;
  const unsigned gx = (per_out + 255u) / 256u;
  unsigned gy = (unsigned)major;
  // enough blocks to fill the chip, but let a thread reuse its tap geometry over several planes when there are many
  // (measured on the small-plane blurs, round 6: 1024 blocks instead of 4096 = more planes per thread: 26 against 23 us)
  while (gy > 1 && (unsigned long long)gx * gy > 4096ull && gy * 2u > (unsigned)major / 32u) gy = (gy + 1) / 2;
  if (gy > 65535u) gy = 65535u;
  const dim3 grid(gx, gy);
  const bool fir4 = kh == 4 && kw == 4 && up_x == up_y && down_x == down_y;
  if (fir4 && up_x == 1 && down_x == 2 && !addend && sizeof(T) == 4 && sizeof(K) == 4 &&
      launch_fir4_down2(reinterpret_cast<float*>(out), reinterpret_cast<const float*>(in),
                        reinterpret_cast<const float*>(kernel), major, in_h, in_w, out_h, out_w, pad_x0, pad_y0, st))
    return gg::launch_status("upfirdn2d_fir4_down2");
  if (fir4 && up_x == 1 && down_x == 2)
    upfirdn2d_direct<T, K, 1><<<grid, 256, 0, st>>>(out, in, kernel, major, in_h, in_w, out_h, out_w, kh, kw, up_x, up_y,
                                                     down_x, down_y, pad_x0, pad_y0, addend);
  else if (fir4 && up_x == 2 && down_x == 1)
    upfirdn2d_direct<T, K, 2><<<grid, 256, 0, st>>>(out, in, kernel, major, in_h, in_w, out_h, out_w, kh, kw, up_x, up_y,
                                                     down_x, down_y, pad_x0, pad_y0, addend);
  else if (fir4 && up_x == 1 && down_x == 1 && fir4_small_fast())
    upfirdn2d_direct<T, K, 3><<<grid, 256, 0, st>>>(out, in, kernel, major, in_h, in_w, out_h, out_w, kh, kw, up_x, up_y,
                                                     down_x, down_y, pad_x0, pad_y0, addend);
  else
    upfirdn2d_direct<T, K, 0><<<grid, 256, 0, st>>>(out, in, kernel, major, in_h, in_w, out_h, out_w, kh, kw, up_x, up_y,
                                                     down_x, down_y, pad_x0, pad_y0, addend);
  return gg::launch_status(what);
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
  const bool blur4 = sizeof(T) == 4 && up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1 && kh == 4 && kw == 4 &&
                     out_h >= 24 && out_w >= 24;
  if (blur4)
    return launch_blur4<0, 0>(reinterpret_cast<float*>(out), reinterpret_cast<const float*>(in),
                                      reinterpret_cast<const float*>(kernel), major, in_h, in_w, out_h, out_w, pad_x0,
                                      pad_y0, BlurFuse{}, st);
  return launch_direct<T, T>(out, in, kernel, nullptr, major, in_h, in_w, out_h, out_w, kh, kw, up_x, up_y, down_x, down_y,
                             pad_x0, pad_y0, st, "upfirdn2d_direct");
}

}  // namespace

extern "C" int gg_upfirdn2d_f32(float* out, const float* in, const float* kernel, int major, int in_h, int in_w,
                                int kernel_h, int kernel_w, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                                int pad_x1, int pad_y0, int pad_y1, void* stream) {
  return upfirdn2d_impl<float>(out, in, kernel, major, in_h, in_w, kernel_h, kernel_w, up_x, up_y, down_x, down_y,
                               pad_x0, pad_x1, pad_y0, pad_y1, stream);
}
// upfirdn2d(in) + addend in one pass: the ToRGB skip connection `rgb + Upsample(skip)` (networks.py:369-371) without
// the separate element-wise add.  addend has the shape of the output.
extern "C" int gg_upfirdn2d_add_f32(float* out, const float* in, const float* kernel, const float* addend, int major,
                                    int in_h, int in_w, int kernel_h, int kernel_w, int up_x, int up_y, int down_x,
                                    int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream) {
  if (up_x < 1 || up_y < 1 || down_x < 1 || down_y < 1 || kernel_h < 1 || kernel_w < 1 || major < 0 || in_h < 0 ||
      in_w < 0)
    return gg::fail(-2, "upfirdn2d_add: bad arguments");
  const int out_h = (in_h * up_y + pad_y0 + pad_y1 - kernel_h + down_y) / down_y;
  const int out_w = (in_w * up_x + pad_x0 + pad_x1 - kernel_w + down_x) / down_x;
  if (out_h <= 0 || out_w <= 0 || major == 0) return 0;
  if (!out || !in || !kernel || !addend) return gg::fail(-2, "upfirdn2d_add: null pointer");
  return launch_direct<float, float>(out, in, kernel, addend, major, in_h, in_w, out_h, out_w, kernel_h, kernel_w, up_x,
                                     up_y, down_x, down_y, pad_x0, pad_y0, gg::as_stream(stream), "upfirdn2d_add");
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
  BlurFuse f;
  f.noise = noise; f.noise_w = noise_weight; f.bias = act_bias; f.ref = ref; f.alpha = alpha; f.gain = gain;
  f.channels = c;
  hipStream_t st = gg::as_stream(stream);
  const long long planes = (long long)n * c;
  if (epi) return launch_blur4<1, 0>(out, in, kernel, planes, in_h, in_w, out_h, out_w, pad_x0, pad_y0, f, st);
  if (pro) return launch_blur4<0, 1>(out, in, kernel, planes, in_h, in_w, out_h, out_w, pad_x0, pad_y0, f, st);
  return launch_blur4<0, 0>(out, in, kernel, planes, in_h, in_w, out_h, out_w, pad_x0, pad_y0, f, st);
}

// gg_blur4_fused_f32 with the activation's sign as a 1-bit plane in the stream kernel's tiling of the FORWARD output
// (H, W) = the backward's input: uint64 [n * c][ceil(H / 16)][ceil(W / 61)][16] (BlurFuse::sign_bits; gg_blur4_bits_words
// uint32 words per plane).  noise != NULL: forward, every word of `bits` is written; noise == NULL: backward, `bits`
// replaces `ref` (bitwise the same result as gg_blur4_fused_f32 on the fp32 output the plane was taken from).
extern "C" int gg_blur4_fused_bits_f32(float* out, const float* in, const float* kernel, int n, int c, int in_h, int in_w,
                                       int pad_x0, int pad_x1, int pad_y0, int pad_y1, const float* noise,
                                       const float* noise_weight, const float* act_bias, unsigned int* bits, float alpha,
                                       float gain, void* stream) {
  const int out_h = in_h + pad_y0 + pad_y1 - 3, out_w = in_w + pad_x0 + pad_x1 - 3;
  if (n <= 0 || c <= 0 || out_h <= 0 || out_w <= 0) return 0;
  if (!out || !in || !kernel || !bits) return gg::fail(-2, "blur4_fused_bits: null pointer");
  const bool epi = noise != nullptr;
  if (epi && (!noise_weight || !act_bias)) return gg::fail(-2, "blur4_fused_bits: noise weight / bias missing");
  BlurFuse f;
  f.noise = noise; f.noise_w = noise_weight; f.bias = act_bias; f.ref = nullptr; f.alpha = alpha; f.gain = gain;
  f.channels = c;
  hipStream_t st = gg::as_stream(stream);
  const long long planes = (long long)n * c;
  // the plane is tiled like the FORWARD output: (H, W) = (out_h, out_w) here in the forward, (in_h, in_w) in the backward
  const int ph = epi ? out_h : in_h, pw = epi ? out_w : in_w;
  f.bit_strips = (pw + SW_OUT - 1) / SW_OUT;
  f.bit_chunks = (ph + SROWS - 1) / SROWS;
  if ((long long)f.bit_strips * f.bit_chunks * SROWS * 8 >= (1LL << 31)) return gg::fail(-2, "blur4_fused_bits: plane too large");
  if (epi) {
    f.sign_bits = reinterpret_cast<unsigned long long*>(bits);
    return launch_blur4<1, 0>(out, in, kernel, planes, in_h, in_w, out_h, out_w, pad_x0, pad_y0, f, st);
  }
  f.ref_bits = bits;
  return launch_blur4<0, 2>(out, in, kernel, planes, in_h, in_w, out_h, out_w, pad_x0, pad_y0, f, st);
}

// dbias[c] (+)= sum over (sample, chunk, strip) of the EPI = 2 launch's strip sums, in a fixed order: thread t of the
// channel's block takes items t, t + 256, ... (ascending), then the fixed tree of block_sum_256
__global__ __launch_bounds__(256) void blur_bias_reduce_kernel(float* __restrict__ dbias, const float* __restrict__ sums,
                                                               int n, int c, int per_plane, int accumulate) {
  __shared__ float red[4];
  const int ch = blockIdx.x;
  const int items = n * per_plane;
  float acc = 0.f;
  for (int i = threadIdx.x; i < items; i += 256) {
    const int s = i / per_plane, u = i - s * per_plane;
    acc += sums[((size_t)s * c + ch) * per_plane + u];
  }
  const float tot = gg::block_sum_256<float>(acc, red);
  if (threadIdx.x == 0) dbias[ch] = (accumulate ? dbias[ch] : 0.f) + tot;
}

// The adjoint of a 4x4 Blur that followed a conv + bias + leaky-ReLU layer, with that activation's backward in its
// epilogue (ResBlock's conv1 -> Blur, networks.py:375-386; fused_act.py:27-38): out = lrelu'(out_ref) * gain *
// blur(in, kernel), kernel = the FLIPPED taps and pads = the adjoint padding (upfirdn2d.py:113-118), out_ref (n, c, out_h,
// out_w) = the layer's saved output; dbias (c), when given, receives (accumulate: is added) the bias gradient
// sum_{n, y, x} out - strip sums in the library's scratch, added per channel in a fixed order.  Replaces the adjoint blur
// followed by gg_fused_lrelu_bwd_f32 (a read of the blurred gradient and of out_ref, a write of the masked gradient).
// GG_NOT_SERVED (nothing launched) for planes below the streaming kernel's 24 x 24 minimum.
extern "C" int gg_blur4_act_bwd_f32(float* out, const float* in, const float* kernel, int n, int c, int in_h, int in_w,
                                    int pad_x0, int pad_x1, int pad_y0, int pad_y1, const float* out_ref, float alpha,
                                    float gain, float* dbias, int accumulate, void* stream) {
  const int out_h = in_h + pad_y0 + pad_y1 - 3, out_w = in_w + pad_x0 + pad_x1 - 3;
  if (n <= 0 || c <= 0 || out_h <= 0 || out_w <= 0) return 0;
  if (!out || !in || !kernel || !out_ref) return gg::fail(-2, "blur4_act_bwd: null pointer");
  if (out_h < 24 || out_w < 24 || blur_use_tile_kernel()) return GG_NOT_SERVED;
  hipStream_t st = gg::as_stream(stream);
  const long long planes = (long long)n * c;
  const int strips_x = (out_w + SW_OUT - 1) / SW_OUT, chunks_y = (out_h + SROWS - 1) / SROWS;
  const long long nunits = planes * strips_x * chunks_y;
  if (nunits >= (1LL << 31) || c > 65535) return gg::fail(-2, "blur4_act_bwd: too many strips / channels");
  BlurFuse f;
  f.noise = nullptr; f.noise_w = nullptr; f.bias = nullptr; f.ref = nullptr; f.alpha = alpha; f.gain = gain;
  f.channels = c;
  f.out_ref = out_ref;
  if (dbias) {
    f.wave_sums = reinterpret_cast<float*>(gg::scratch(st, sizeof(float) * (size_t)nunits));
    if (!f.wave_sums) return -3;
  }
  const int rc = launch_blur4<2, 0>(out, in, kernel, planes, in_h, in_w, out_h, out_w, pad_x0, pad_y0, f, st);
  if (rc || !dbias) return rc;
  blur_bias_reduce_kernel<<<(unsigned)c, 256, 0, st>>>(dbias, f.wave_sums, n, c, strips_x * chunks_y, accumulate ? 1 : 0);
  return gg::launch_status("blur_bias_reduce");
}

// words (uint32) per plane of gg_blur4_fused_bits_f32's sign plane for an (h, w) forward output
extern "C" int gg_blur4_bits_words(int h, int w) {
  if (h <= 0 || w <= 0) return 0;
  const long long words = 2LL * ((w + SW_OUT - 1) / SW_OUT) * ((h + SROWS - 1) / SROWS) * SROWS;
  return words < (1LL << 29) ? (int)words : -1;
}
// binary16 tensors (the reference dispatches half, upfirdn2d_kernel.cu:311); taps stay fp32
extern "C" int gg_upfirdn2d_f16(unsigned short* out, const unsigned short* in, const float* kernel, int major, int in_h,
                                int in_w, int kernel_h, int kernel_w, int up_x, int up_y, int down_x, int down_y,
                                int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream) {
  if (up_x < 1 || up_y < 1 || down_x < 1 || down_y < 1 || kernel_h < 1 || kernel_w < 1 || major < 0 || in_h < 0 ||
      in_w < 0)
    return gg::fail(-2, "upfirdn2d: bad arguments");
  const int out_h = (in_h * up_y + pad_y0 + pad_y1 - kernel_h + down_y) / down_y;
  const int out_w = (in_w * up_x + pad_x0 + pad_x1 - kernel_w + down_x) / down_x;
  if (out_h <= 0 || out_w <= 0 || major == 0) return 0;
  if (!out || !in || !kernel) return gg::fail(-2, "upfirdn2d: null pointer");
  return launch_direct<_Float16, float>(reinterpret_cast<_Float16*>(out), reinterpret_cast<const _Float16*>(in), kernel,
                                        nullptr, major, in_h, in_w, out_h, out_w, kernel_h, kernel_w, up_x, up_y, down_x,
                                        down_y, pad_x0, pad_y0, gg::as_stream(stream), "upfirdn2d_direct");
}
extern "C" int gg_upfirdn2d_f64(double* out, const double* in, const double* kernel, int major, int in_h, int in_w,
                                int kernel_h, int kernel_w, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                                int pad_x1, int pad_y0, int pad_y1, void* stream) {
  return upfirdn2d_impl<double>(out, in, kernel, major, in_h, in_w, kernel_h, kernel_w, up_x, up_y, down_x, down_y,
                                pad_x0, pad_x1, pad_y0, pad_y1, stream);
}
