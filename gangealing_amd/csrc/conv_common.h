// Types and device helpers shared by the convolution translation units (conv_mfma.hip, conv_s2_patch.hip):
// the launch descriptor, raw buffer access, the limb formats of the split-precision kernels and the block exponent
// of the binary16 limbs.  Everything here is a type, a constant or a force-inlined device function.
#pragma once
#include "../../include/gangealing_hip.h"
#include "gg_common.h"

namespace gg_conv {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;

struct ConvArgs {
  float* y;
  const float* x;
  const float* wmat;
  const float* in_scale;
  const float* out_scale;
  const float* bias;
  int batch, groups, cin_g, cout_g, h, w, oh, ow;
  int mh, mw;            // M-space grid per image (pixels of this launch)
  int ys, yo, xs, xo;    // output coordinate = q * s + o
  int bs, byo, bxo;      // gather base coordinate = q * bs + bo
  int py, px;            // parity class (MODE 1)
  int nty, ntx;          // taps per axis of this class (MODE 1); MODE 0: KS
  int ktot;              // cin_g * ntaps
  int tiles_co, tiles_pix;
  int splitk, slabs_per_split, nslabs;
  int tile_pixels;                // pixels per block tile chosen by plan_conv
  const unsigned short* wsplit;   // bf16 limb planes [limb][g][co][k = (tap, ci)]  (split-precision path)
  long long wsplit_stride;        // elements between limb planes
  // optional StyledConv tail fused into the epilogue: y = lrelu(acc + noise_w[0]*noise[n,pix] + act_bias[co]) * gain
  // optional leaky-ReLU gradient mask on the INPUT (data gradient of a conv + activation layer): the gathered
  // element x is multiplied by (mask_ref > 0 ? 1 : mask_alpha) * mask_gain, mask_ref = the layer's saved output
  const float* mask_ref;
  float mask_alpha, mask_gain;
  // round 5: the same mask from ONE BIT per element instead of the saved fp32 output.  sign_bits (forward launches with
  // the fused activation): the 3x3 patch tile's epilogue additionally writes bit (co & 31) of word
  // [(n * H * W + pixel) * bit_words + co / 32] = (stored output > 0), bit_words = cout / 32.  mask_bits (data-gradient
  // launches): that plane in place of mask_ref - a thread of the patch gather (= one patch pixel, the chunk's 32
  // channels) reads one word where it read 32 floats.
  unsigned* sign_bits;
  const unsigned* mask_bits;
  int bit_words;
  int act;                        // 1: fused bias / noise / leaky-ReLU epilogue
  const float* act_noise;         // (N, 1, OH, OW) or null = no noise term
  const float* act_noise_w;       // device scalar
  const float* act_bias;          // (groups*cout_g)
  float act_alpha, act_gain;
  // split-K launches (and launches that share an output with one): instead of y, block (tile, split) writes its RAW
  // accumulators to part + split * part_stride (same element offsets as y); splitk_reduce_kernel then adds the
  // splits in ascending order and applies out_scale / bias / activation.  A fixed summation order: results are
  // bitwise reproducible (float atomics onto y would combine the splits in arrival order).
  float* part;
  long long part_stride;
  int f16;                        // split-precision limbs are binary16 (forward convolutions), see Limb<>
  float acc_scale;                // accumulators are multiplied by this first (1 / kF16WeightScale with f16 limbs)
  int nt_store;                   // the 8-wave tiles' epilogues store with the non-temporal hint (outputs beyond the caches)
  // binary16 limbs: lower edge of the E = 0 band of the block exponent (see f16_block_exp).  2^-3 for forward operands
  // (activations: bit-identical to the unscaled kernel over their usual range), 2^5 for GRADIENT operands (format code
  // bit 5): a chunk is then staged unscaled only where that loses nothing against the normalised form
  float exp_lo;
  // MEASUREMENT ONLY (gg_debug_set_prelimb, scripts/prelimb_probe.py): the operand x already in MFMA-ready form - style
  // applied, two binary16 limbs, [image][16-channel chunk][pixel][limb][16 channels] (64 B per pixel and chunk, the LDS
  // rows of conv_t_c16.hip) - what a producer epilogue would write.  Null in every product launch.
  const unsigned short* xlimb;
  // round 6, production form of the same: xlimb_e[image] = the power-of-two exponent the limbs were scaled by (the consumer's
  // block exponent, one per image: 2^-E applied by the producer pass, 2^E by this kernel's epilogue)
  const int* xlimb_e;
  // forward launches with the fused activation: amax_out[image] receives max |stored output| (atomic max on the bit
  // pattern of a non-negative float: order-independent, so reproducible) - what the limb-writing pass takes its exponent from
  float* amax_out;
  // round 6 (second half): a tensor of the OUTPUT's shape added after scale / bias (/ activation) - ResBlock's residual
  // merge inside the skip branch's 1x1 convolution (networks.py:392-393).  Carried by the generic tiles' epilogues and by
  // the split-K reduce pass only (1x1 convolutions always take those).
  const float* residual;
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BKS = 32;                 // K per slab
constexpr int EPT = BKS / 2;            // gathered elements (and weight k's) per thread per slab
constexpr int ROWB = BKS * 2 + 16;      // bytes per LDS row

typedef unsigned U4 __attribute__((ext_vector_type(4)));

// raw buffer load of one float: address = resource base + voffset (per lane) + soffset (scalar); an offset at or
// beyond the resource's num_records returns 0.  kOobOffset selects that for masked lanes (resources are < 2 GiB).
constexpr unsigned kOobOffset = 0x80000000u;
// The descriptor inputs go through readfirstlane so that the compiler can PROVE they are wave-uniform; otherwise
// every buffer op is wrapped in a waterfall loop.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const float* p, int bytes) {
  const unsigned long long u = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                           __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ U4 buffer_load_u4(__amdgpu_buffer_rsrc_t r, unsigned voffset, int soffset) {
  return __builtin_bit_cast(U4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voffset, soffset, 0));
}
__device__ __forceinline__ float buffer_load_f32(__amdgpu_buffer_rsrc_t r, unsigned voffset, int soffset) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voffset, soffset, 0));
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
// stores need dword alignment only; an offset beyond num_records makes the store vanish
__device__ __forceinline__ void buffer_store_f32x4(f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voffset, int soffset) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(U4, v), r, (int)voffset, soffset, 0);
}
__device__ __forceinline__ void buffer_store_f32(float v, __amdgpu_buffer_rsrc_t r, unsigned voffset, int soffset) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, (int)voffset, soffset, 0);
}
// the same with the non-temporal hint (aux = 2): for outputs far larger than L2 + Infinity Cache, which their consumer
// will read from HBM anyway; measured on the 541 MB up-convolution output: -5.6 % of the launch, -1 % on the 268 - 537 MB
// stride-1 outputs, +4 % on a 138 MB output that the next kernel would have found in the Infinity Cache - hence per
// launch (ConvArgs::nt_store; profiles/r03_s_nt_store_experiment.txt)
__device__ __forceinline__ void buffer_store_f32x4_nt(f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voffset, int soffset) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(U4, v), r, (int)voffset, soffset, 2);
}
constexpr long long kNtStoreBytes = 256LL << 20;          // Infinity Cache size
// orders one wave's LDS writes before its own later LDS reads (and vice versa) when the lanes exchange data through a
// region no other wave touches: DS operations of a wave execute in order, so only the compiler has to be held back
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}   // a native vector: a struct here is kept in scratch by the compiler

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v;
  v[0] = (__bf16)a;
  v[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf16_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf16_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// Limb format of the split-precision kernels.  F16 = false: bf16 limbs (8-bit mantissa, fp32's exponent range - any
// operand, in particular gradients of arbitrary magnitude).  F16 = true: IEEE binary16 limbs (11-bit mantissa): two
// limbs carry 22 bits, so the same THREE MFMA products (x0 w0 + x0 w1 + x1 w0) leave ~2^-21 per product instead of
// ~2^-17 - fp32-class results at the two-limb price - but binary16 spans only 6e-8 .. 65504, so this format is used
// where the operand range is known: the FORWARD convolutions (activations of O(1); weights are pre-scaled by
// kF16WeightScale in the pack and the accumulators multiplied by its inverse, so that the low weight limb stays a
// normal number).  Values beyond +-65504 saturate instead of becoming inf.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float kF16WeightScale = 256.f;
template <bool F16>
struct Limb;
// pack2 / one: `first` = this is limb 0 of the value (the others hold residuals, which are small)
template <>
struct Limb<false> {
  static __device__ __forceinline__ unsigned pack2(float a, float b, bool) { return pack_bf16x2(a, b); }
  static __device__ __forceinline__ float lo(unsigned p) { return bf16_lo(p); }
  static __device__ __forceinline__ float hi(unsigned p) { return bf16_hi(p); }
  static __device__ __forceinline__ unsigned short one(float v, bool) { return __builtin_bit_cast(unsigned short, (__bf16)v); }
  static __device__ __forceinline__ float back(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
  static __device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <>
struct Limb<true> {
  // Limb 0 is converted with round-toward-zero (v_cvt_pkrtz_f16_f32: one instruction per pair, and under that rounding
  // an overflow yields +-65504 instead of inf - saturation for free); its residual (< one binary16 ulp of the value,
  // exactly representable in fp32) goes to limb 1 with round-to-nearest, so what is finally dropped is <= 2^-22 |x|
  // and unbiased, exactly as with two nearest roundings.
  static __device__ __forceinline__ unsigned pack2(float a, float b, bool first) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    if (first) return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
    f16x2 v;
    v[0] = (_Float16)a;
    v[1] = (_Float16)b;
    return __builtin_bit_cast(unsigned, v);
  }
  static __device__ __forceinline__ float lo(unsigned p) {
    return (float)__builtin_bit_cast(_Float16, (unsigned short)(p & 0xffffu));
  }
  static __device__ __forceinline__ float hi(unsigned p) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(p >> 16)); }
  static __device__ __forceinline__ unsigned short one(float v, bool first) {
    return (unsigned short)(pack2(v, 0.f, first) & 0xffffu);
  }
  static __device__ __forceinline__ float back(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }
  static __device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};
// ---- block floating point for the binary16 limbs (round 4) ---------------------------------------------------------
// binary16 spans 6e-8 .. 65504, fp32 operands do not: activation x style reaches 1e5 in trained generators, gradients
// sit at 1e-8.  Every block therefore carries ONE power-of-two exponent E for the operand it stages (activations or
// gradients, after the style / mask factors): values are multiplied by 2^-E before they are split into limbs and the
// accumulators by 2^E in the epilogue - both exact.  E comes from the data itself: while a chunk (32 input channels of
// the tile's patch, or one gathered slab) waits in registers, the block takes the maximum magnitude of what it is about
// to stage (v_max per element, one wave butterfly, one LDS word per wave, published by a barrier the loop already has).
//   * amax in [lo, 2^11]: E = 0, nothing is scaled (16x headroom to 65504).  lo = 2^-3 for forward operands (activations:
//     bit-identical to the unscaled kernel over their usual range; an element keeps an ABSOLUTE 2^-25, which the 1e-4
//     activation bound never sees).  Round 6: lo = 2^5 for GRADIENT operands (ConvArgs::exp_lo).  With 2^-3 a gradient
//     chunk holding one outlier in [0.125, 1) among entries of 1e-5 was staged unscaled and its small entries kept only
//     2^-25 / 1e-5 = 3e-3 relative - 512x worse than the same chunk an ulp below the band edge, invisible to tests that
//     bound err / max|ref|.  At 2^5 the unscaled form's floor (2^-25 <= 2^-30 amax) is what normalising would give;
//   * otherwise E = floor(log2 amax) - 6, i.e. amax 2^-E in [2^6, 2^7): limb 0 keeps 11 bits of every element down to
//     2^-14 of that, limb 1 another 11 bits down to 2^-3 and an ABSOLUTE 2^-25 below - 2^-31 of the chunk's largest
//     element, so the error of a dot product is 2^-22 of its terms' scale whatever the operand's magnitude;
//   * E only grows inside a tile: when a later chunk needs a larger exponent the accumulators are multiplied by
//     2^(E_old - E_new) once (exact) and E moves; a chunk that is much smaller than its predecessors is staged with the
//     tile's E (its terms are small against the sum already accumulated).
// No state outside the block, no calibration pass, no saturation: overflow cannot happen for finite inputs, and two
// runs of the same launch do the same arithmetic (the maximum does not depend on the order of its operands).
struct BlockExp {
  int e = 0;          // current exponent
  int set = 0;        // a non-zero chunk has been seen
};
// Exponent range: a finite fp32 amax has floor(log2) in [-149, 127], so E = floor(log2 amax) - 6 lies in [-155, 121]; it is
// clamped to [-120, 121] so that 2^-E and 2^E are normal fp32 numbers (round 4 clamped to +-100: operands above 2^111 =
// 2.6e33 saturated limb 0).  Below 2^-120 * 2^6 the operand keeps fewer than 22 bits - of values that are denormal or
// within 2^6 of it.  exp2i saturates outside [-126, 127]: a rescale factor 2^(E_old - E_new) below 2^-126 (a tile whose
// chunks differ by more than 126 binades) flushes the older, negligible, accumulators to zero instead of producing
// garbage bits.
__device__ __forceinline__ int f16_block_exp(float amax, float lo) {          // amax > 0, uniform
  if (amax >= lo && amax <= 2048.f) return 0;
  int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu) - 127 - 6;
  return e < -120 ? -120 : (e > 121 ? 121 : e);
}
__device__ __forceinline__ float exp2i(int e) {
  if (e < -126) return 0.f;
  if (e > 127) e = 127;
  return __builtin_bit_cast(float, (unsigned)(127 + e) << 23);
}
// -> factor for the accumulators (1 = leave them), updates `b` for a chunk whose largest magnitude is `amax`
__device__ __forceinline__ float block_exp_update(BlockExp& b, float amax, float lo) {
  if (!(amax > 0.f)) return 1.f;
  if (!b.set) {
    b.set = 1;
    b.e = f16_block_exp(amax, lo);
    return 1.f;                       // the accumulators are still zero
  }
  if (amax * exp2i(-b.e) <= 2048.f) return 1.f;
  const int ne = f16_block_exp(amax, lo);
  const float f = exp2i(b.e - ne);    // ne > b.e
  b.e = ne;
  return f;
}
// maximum over the wave, then lane 0 publishes it in slot[wave]; the caller's next barrier makes it visible
// (DPP, not ds_bpermute shuffles: six VALU instructions and no LDS round trips - the reduction sits inside the MFMA
// stream of the tap loop)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max(float v) {
  const int iv = __builtin_bit_cast(int, v);
  return fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, CTRL, ROW_MASK, 0xf, false)));
}
__device__ __forceinline__ void publish_wave_amax(float m, float* slot, int wid, int lane) {
  m = dpp_max<0x128, 0xf>(m);        // row_ror:8, 4, 2, 1: every lane of a 16-lane row holds the row's maximum
  m = dpp_max<0x124, 0xf>(m);
  m = dpp_max<0x122, 0xf>(m);
  m = dpp_max<0x121, 0xf>(m);
  m = dpp_max<0x142, 0xa>(m);        // row_bcast:15 into rows 1 and 3
  m = dpp_max<0x143, 0xc>(m);        // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's maximum
  if (lane == 63) slot[wid] = m;
}
template <int NW>
__device__ __forceinline__ float read_block_amax(const float* slot) {
  float m = slot[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) m = fmaxf(m, slot[i]);
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m)));
}

// host-side-exact variant for the weight packs (run once per weight version): nearest rounding with explicit saturation
__device__ __forceinline__ unsigned short f16_limb_rn(float v) {
  return __builtin_bit_cast(unsigned short, (_Float16)fminf(fmaxf(v, -65504.f), 65504.f));
}

// launch descriptor of the weight-gradient kernels
struct WgradArgs {
  float* dw;
  const float* x;
  const float* dy;
  int batch, groups, cin_g, cout_g, h, w, oh, ow, stride, pad;
  int jtot;                     // cin_g * KS*KS
  int tiles_co, tiles_j;
  long long ktot;               // batch*oh*ow
  long long k_per_split;
  float scale;
  // optional (row-streaming kernel): dy is the gradient w.r.t. a leaky-ReLU OUTPUT; the activation's backward
  // dy * (mask_ref > 0 ? 1 : mask_alpha) * mask_gain is applied while dy is staged, and its per-channel sum (the
  // bias gradient) is accumulated into dbias
  const float* mask_ref;
  float mask_alpha, mask_gain;
  float* dbias;
  // generic kernels: K-splits write their raw tiles to part[(split * groups + g) * cout_g * jtot + ...] (summed in
  // split order by a reduce pass - no float atomics); null: a single split stores (or adds, `accumulate`) into dw
  float* part;
  int accumulate;
};

// conv_s2_wgrad.hip: 3x3 / stride 2 / pad 0 weight gradient, row-streaming (workspace layout and reduce pass of
// conv3x3_wgrad_rows_kernel); tco x tci = 128 x 32 or 64 x 64
void s2_wgrad_rows_launch(const WgradArgs& a, int limbs, bool narrow, int segs, int rblocks, int rows_per_block,
                          int units_per_block, float* ws, dim3 grid, hipStream_t st);

// conv_s2_patch.hip: 3x3 / stride 2 / pad 0 correlation with input-patch reuse (tpix = 128 or 256 output pixels per
// block, 128 output channels; a.tiles_co / tiles_pix / nslabs (16-channel chunks) / slabs_per_split / part set by the caller)
constexpr int kS2MaxCin = 1024;            // the image-group's in_scale vector is kept in LDS
bool s2_patch_serves(const ConvArgs& a, int tpix);
void s2_patch_launch(const ConvArgs& a, int stride, int tpix, dim3 grid, hipStream_t st);

// conv_t_c16.hip: 3x3 / stride 2 transposed convolution, all four parity classes per pass, 16-channel chunks (two
// limbs; tco = 64: four waves, two blocks per CU; 128: eight waves); geometry arguments as convT3x3s2_patch_kernel's,
// a.tiles_co / tiles_pix / nslabs (16-channel chunks) / slabs_per_split / part set by the caller
bool t16_serves(const ConvArgs& a);
void t16_launch(const ConvArgs& a, int tco, int tw_log2, int tiles_y, int edge, int pad, dim3 grid, hipStream_t st);
void t16_limb_convert(unsigned short* out, const float* x, const float* scale, int planes, int hw, hipStream_t st);
// ToRGB (3 outputs) + limb form of the same activation for the next up-convolution; -1: shape not served
int t16_torgb_limb(float* rgb, unsigned short* xlimb, int* xexp, const float* x, const float* wmat,
                   const float* rgb_style, const float* bias, const float* next_style, const float* amax, int batch,
                   int cin, long long hw, hipStream_t st);

}  // namespace gg_conv
