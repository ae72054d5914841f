/*
 * gangealing_hip.h - C ABI of the MI355X (gfx950) GANgealing hot-path library
 * (libgangealing_hip.so, built from gangealing_amd/csrc/ with hipcc --offload-arch=gfx950).
 *
 * This is the drop-in boundary.  Every entry point takes plain device pointers, sizes and a
 * hipStream_t passed as void* (0 = the null stream); there are no torch types in any signature.
 * Each function cites the reference interface it replaces (paths relative to wpeebles/gangealing).
 *
 * Conventions
 *   - return value: 0 on success; otherwise a hipError_t value (> 0) or a negative argument-error
 *     code.  gg_last_error() returns a thread-local, NUL-terminated description.  The optional fused
 *     entry points gg_conv3x3_masked_dgrad_f32 / gg_conv3x3_masked_wgrad_f32 return GG_NOT_SERVED when
 *     their kernel does not cover the shape: nothing was launched and the caller uses the unfused pair.
 *   - all tensors are dense row-major ("contiguous" in torch terms); NCHW unless noted.
 *   - kernels are enqueued on `stream` and never synchronise.
 *   - reproducibility: no floating-point atomics on the training path.  Split reductions (split-K convolutions,
 *     K-split weight gradients, grid-wide sums, scatter-shaped gradients) keep their partial results in a
 *     per-(device, stream) SCRATCH buffer owned by the library and add them in a fixed order, so results are
 *     bitwise identical run to run.  The scratch grows on demand (a few times, during the first calls at the
 *     largest shapes) out of the allocator installed with gg_set_allocator (ABI 3; a caching allocator may serve it
 *     during a hipGraph capture) or, without one, with hipMalloc - which is not legal while a stream is being
 *     captured.  The stream's TICKET PAGE (zero-initialised arrival counters of the grid-wide sums) is different: it
 *     must exist before a capture begins - its clearing memset would only be recorded - so the entry points fail
 *     (code -4) when its first use falls inside a capture: run the step eagerly once on that stream, or call
 *     gg_scratch_reserve(0, stream), before capturing.  Exceptions to the reproducibility rule, documented at the entry
 *     points: the IMAGE gradient of the warp (gg_mipmap_warp_bwd_f32 with grad_pyr*, gg_mip_downsample2x_bwd_f32) and
 *     gg_splat_forward_f32 / gg_splat2d_f32 for points whose box exceeds 33 pixels scatter with float atomics -
 *     neither is on the training path.
 *   - outputs are fully overwritten unless the comment says "accumulates".
 */
#ifndef GANGEALING_HIP_H
#define GANGEALING_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define GG_NOT_SERVED (-1000)

/* 2: library-owned scratch (gg_scratch_*), gg_lpips_tail_bwd_f32 gained `accumulate`
 * 3: gg_set_allocator; binary16 limbs (format code 18) accepted by the data-gradient entry points
 * 4: gg_modconv3x3_act_bits_f32 / gg_conv3x3_masked_dgrad_bits_f32 (1-bit sign plane), gg_set_tuning; the ticket page of
 *    a stream may not be created inside a hipGraph capture (error -4)
 * 5: gg_mipmap_warp_fwd/bwd_f32 take the pyramid as (level 0, the deeper levels in one buffer, num_levels <= 8) instead of
 *    four level pointers; SplatForwardGpu (the reference's own symbol) exported
 * 6: additive - gg_similarity_matrix_f32 / _bwd_f32, gg_conv3x3_fewout_masked_bits_f32 (+ the few-input-channel stem kernel
 *    behind gg_modconv3x3_act_bits_f32 with limbs = 0), gg_blur4_fused_bits_f32 / gg_blur4_bits_words,
 *    gg_conv2d_split_act_f32, gg_conv1x1_split_residual_f32, gg_blur4_act_bwd_f32 */
int gg_abi_version(void);
/* Pre-size the scratch buffer of `stream` on the current device to at least `bytes` and create its ticket page.
 * Optional for eager use (the entry points grow the scratch on demand); REQUIRED once per stream before a hipGraph
 * capture whose kernels take tickets, unless the step already ran eagerly on that stream.  Not legal during a capture. */
int gg_scratch_reserve(long long bytes, void* stream);
/* Free every scratch buffer (synchronises the device).  Graphs captured earlier must not be replayed afterwards. */
int gg_scratch_release(void);
/* ABI 3: source of the scratch memory.  By default the library calls hipMalloc / hipFree.  A host that runs a caching
 * device allocator installs it here BEFORE the first scratch use: `alloc(bytes)` returns device memory valid on the
 * current device and usable on the calling thread's current stream (NULL on failure), `free(ptr)` takes it back.
 * gangealing_amd/_lib.py installs torch's allocator: the scratch is then part of torch's pool (no allocation outside
 * its accounting, and growth during a hipGraph capture is legal).  Pass two NULLs to restore the default. */
typedef void* (*gg_alloc_fn)(long long bytes);
typedef void (*gg_free_fn)(void* ptr);
int gg_set_allocator(gg_alloc_fn alloc, gg_free_fn free_fn);
/* Name of the kernel instantiation the calling thread's last convolution entry point launched (tile shape, limb format;
 * "" before the first call): measurement aid - bench.py keys its per-kernel HIP-event timing on it. */
const char* gg_last_conv_kernel(void);
/* Measurement aid: override a kernel-selection switch of the convolution dispatcher at run time (same names and values
 * as the GG_* environment switches it reads at first use: "GG_CONVT16" 0 | 1 | 64 | 128, "GG_CONVT16_TW").  Results do
 * not depend on these switches beyond rounding order; tests use it to run the same shapes through every tile.
 * value = GG_TUNING_RESET restores the environment's / built-in setting. */
#define GG_TUNING_RESET (-2147483647 - 1)
int gg_set_tuning(const char* name, int value);
const char* gg_last_error(void);
/* Name of the gfx target the device code was built for ("gfx950"). */
const char* gg_build_arch(void);

/* ------------------------------------------------------------------------------------------
 * a2  fused bias + activation.
 * Replaces fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale)
 *   (models/stylegan2/op/fused_bias_act.cpp:11-21, fused_bias_act_kernel.cu:18-99):
 *   v = x[i] + bias[(i / step_b) % size_b];  y = f(v) * scale, f selected by act*10+grad:
 *   act 1 = linear, act 3 = leaky relu with slope alpha; grad 1 uses `ref` as sign reference,
 *   grad 2 yields 0.  bias / ref may be NULL ("empty tensor" in the reference).
 * ------------------------------------------------------------------------------------------ */
int gg_fused_bias_act_f32(float* out, const float* x, const float* bias, const float* ref,
                          int act, int grad, float alpha, float scale,
                          long long size_x, long long step_b, int size_b, void* stream);
int gg_fused_bias_act_f64(double* out, const double* x, const double* bias, const double* ref,
                          int act, int grad, double alpha, double scale,
                          long long size_x, long long step_b, int size_b, void* stream);
/* IEEE binary16 tensors passed as their 16-bit patterns (the reference dispatches half:
 * fused_bias_act_kernel.cu:89); fp32 arithmetic inside, one rounding per element. */
int gg_fused_bias_act_f16(unsigned short* out, const unsigned short* x, const unsigned short* bias,
                          const unsigned short* ref, int act, int grad, float alpha, float scale,
                          long long size_x, long long step_b, int size_b, void* stream);

/* Backward of fused_leaky_relu in ONE pass: grad_in = (out > 0 ? g : alpha*g) * scale and
 * grad_bias[c] = sum_{n,hw} grad_in (the reference re-reads grad_in in a second torch reduction,
 * models/stylegan2/op/fused_act.py:22-40).  Tensors are (n, c, hw) dense; grad_bias (c,) is
 * overwritten.  grad_bias may be NULL to skip the reduction. */
int gg_fused_lrelu_bwd_f32(float* grad_in, float* grad_bias, const float* grad_out, const float* out,
                           float alpha, float scale, int n, int c, long long hw, void* stream);
/* as gg_fused_lrelu_bwd_f32; accumulate != 0: grad_bias[c] += the sum (gradient arenas: no temporary, no extra add) */
int gg_fused_lrelu_bwd_acc_f32(float* grad_in, float* grad_bias, const float* grad_out, const float* out,
                               float alpha, float scale, int n, int c, long long hw, int accumulate, void* stream);
int gg_fused_lrelu_bwd_f64(double* grad_in, double* grad_bias, const double* grad_out, const double* out,
                           double alpha, double scale, int n, int c, long long hw, void* stream);
/* binary16 tensors; grad_bias is an fp32 buffer of c entries (sum of the ROUNDED grad_in values, as the
 * reference's grad_input.sum does) */
int gg_fused_lrelu_bwd_f16(unsigned short* grad_in, float* grad_bias, const unsigned short* grad_out,
                           const unsigned short* out, float alpha, float scale, int n, int c, long long hw,
                           void* stream);

/* StyledConv tail (networks.py:291-298,344-350) in one pass:
 *   out = lrelu(x + noise_weight[0] * noise[n,0,hw] + bias[c], alpha) * scale
 * x/out (n,c,hw), noise (n,1,hw), noise_weight: device scalar (NoiseInjection.weight), hw % 4 == 0.
 * The backward is gg_fused_lrelu_bwd (the sign reference is `out`, as for fused_leaky_relu). */
int gg_noise_bias_act_f32(float* out, const float* x, const float* noise, const float* noise_weight,
                          const float* bias, float alpha, float scale, int n, int c, long long hw, void* stream);

/* ------------------------------------------------------------------------------------------
 * a1  upfirdn2d.
 * Replaces upfirdn2d_op.upfirdn2d(input[M,H,W,1], kernel[kh,kw], up_x, up_y, down_x, down_y,
 *   pad_x0, pad_x1, pad_y0, pad_y1) -> [M,out_h,out_w,1]
 *   (models/stylegan2/op/upfirdn2d.cpp:12-23, upfirdn2d_kernel.cu:209-368), minor_dim == 1
 *   (always, on this path: upfirdn2d.py:101).  out_h = (in_h*up_y + pad_y0 + pad_y1 - kh)/down_y + 1.
 * `out` must hold major * out_h * out_w elements.  The backward pass is the same entry point with
 * up<->down, flipped taps and g_pad (upfirdn2d.py:21-62).
 * ------------------------------------------------------------------------------------------ */
int gg_upfirdn2d_f32(float* out, const float* in, const float* kernel,
                     int major, int in_h, int in_w, int kernel_h, int kernel_w,
                     int up_x, int up_y, int down_x, int down_y,
                     int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);
int gg_upfirdn2d_f64(double* out, const double* in, const double* kernel,
                     int major, int in_h, int in_w, int kernel_h, int kernel_w,
                     int up_x, int up_y, int down_x, int down_y,
                     int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);
/* binary16 tensors (the reference dispatches half: upfirdn2d_kernel.cu:311), fp32 taps and accumulation */
int gg_upfirdn2d_f16(unsigned short* out, const unsigned short* in, const float* kernel,
                     int major, int in_h, int in_w, int kernel_h, int kernel_w,
                     int up_x, int up_y, int down_x, int down_y,
                     int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);
/* upfirdn2d(in) + addend (addend shaped like the output) in one pass: ToRGB's `out + self.upsample(skip)`
 * (networks.py:369-371) without the separate element-wise add. */
int gg_upfirdn2d_add_f32(float* out, const float* in, const float* kernel, const float* addend, int major, int in_h,
                         int in_w, int kernel_h, int kernel_w, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                         int pad_x1, int pad_y0, int pad_y1, void* stream);
/* out = (a + b) * scale over n floats: ResBlock's residual merge (out + skip) / sqrt(2) (networks.py:392-393). */
int gg_add_scale_f32(float* out, const float* a, const float* b, float scale, long long n, void* stream);

/* 4x4 FIR blur (up = down = 1; upfirdn2d.py:147-158 with a 4x4 kernel) with the neighbouring element-wise
 * stage of the generator's up-sampling StyledConv fused in (networks.py:268-298, 344-350):
 *   noise != NULL: out = lrelu(blur(in) + noise_weight[0] * noise[n,0] + act_bias[c], alpha) * gain   (forward)
 *   ref   != NULL: out = blur(in * (ref > 0 ? 1 : alpha) * gain)      (backward: `kernel` already flipped, the
 *                  adjoint padding of upfirdn2d.py:113-118, ref = the saved forward output)
 *   neither: the plain blur.  in (n,c,in_h,in_w), out (n,c,in_h+pad_y0+pad_y1-3, in_w+pad_x0+pad_x1-3). */
int gg_blur4_fused_f32(float* out, const float* in, const float* kernel, int n, int c, int in_h, int in_w,
                       int pad_x0, int pad_x1, int pad_y0, int pad_y1, const float* noise,
                       const float* noise_weight, const float* act_bias, const float* ref, float alpha, float gain,
                       void* stream);
/* Round 6 (second half): the same with the activation's sign as ONE BIT per element, in the kernel's own tiling of the
 * forward's output (H, W) = the backward's input: bits = uint64 [n * c][ceil(H / 16)][ceil(W / 61)][16 rows], bit j of a
 * word = column strip * 61 + j of that row (gg_blur4_bits_words(H, W) uint32 words per plane; 8-byte aligned).
 *   noise != NULL (forward): every word of `bits` is written with (out > 0) - no atomics, nothing to clear;
 *   noise == NULL (backward): out = blur(in * (bit ? 1 : alpha) * gain) - bitwise what gg_blur4_fused_f32 gives on the
 *   fp32 output the plane was taken from, at a third less traffic (the mask stream is ~1 / 25 of its size). */
int gg_blur4_bits_words(int h, int w);
/* Round 6 (second half): the ADJOINT of a 4x4 Blur that followed a conv + bias + leaky-ReLU layer (ResBlock's conv1 -> Blur,
 * networks.py:375-386) with that activation's backward (fused_act.py:27-38) in its epilogue:
 *   out = (out_ref > 0 ? 1 : alpha) * gain * blur(in, kernel)      kernel = the flipped taps, pads = the adjoint padding
 *   dbias[c] (=, or += with accumulate) sum_{n,y,x} out            (NULL: not wanted) - strip sums added in a fixed order
 * in (n,c,in_h,in_w) = the gradient of the blur's output, out / out_ref (n,c,in_h+pad_y0+pad_y1-3, ...) = the layer's
 * output.  Replaces gg_upfirdn2d_f32 + gg_fused_lrelu_bwd(_acc)_f32.  GG_NOT_SERVED below 24 x 24 outputs. */
int gg_blur4_act_bwd_f32(float* out, const float* in, const float* kernel, int n, int c, int in_h, int in_w, int pad_x0,
                         int pad_x1, int pad_y0, int pad_y1, const float* out_ref, float alpha, float gain, float* dbias,
                         int accumulate, void* stream);
int gg_blur4_fused_bits_f32(float* out, const float* in, const float* kernel, int n, int c, int in_h, int in_w,
                            int pad_x0, int pad_x1, int pad_y0, int pad_y1, const float* noise,
                            const float* noise_weight, const float* act_bias, unsigned int* bits, float alpha, float gain,
                            void* stream);

/* ------------------------------------------------------------------------------------------
 * a11  splat2d.
 * gg_splat_forward_f32 replaces
 *   extern "C" void SplatForwardGpu(stream, coords, values, sigma, alpha_splats, output,
 *                                   num_points, channels, height, width, top_count)
 *   (utils/splat2d_cuda/src/splat_gpu_impl.cuh:11-22, splat_gpu_impl.cu:41-122): accumulates
 *   Gaussian weights into alpha_splats (N,H,W) and weight*value into output (N,C,H,W).
 * gg_splat2d_f32 replaces splat_forward_cuda (splat_gpu.c:12-42): output = clone(input) + splats,
 *   then output /= ((soft_normalize ? max(alpha,1) : alpha) + 1e-8).  alpha_ws is an (N,H,W)
 *   caller-provided workspace.
 * Round 4: points whose +-2 sigma box is at most 33 pixels wide (sigma <= 8) are binned per 32 x 32-pixel tile (lists in
 *   the library's scratch, sorted by point index) and gathered by one block per tile - no floating-point atomics, results
 *   bitwise reproducible, every pixel written once (gg_splat2d_f32 fuses the normalisation into that write).  Larger
 *   boxes use hardware float atomics (arrival order), as the reference kernel does for every point.
 * ------------------------------------------------------------------------------------------ */
int gg_splat_forward_f32(const float* coords, const float* values, const float* sigma,
                         float* alpha_splats, float* output,
                         int num_points, int channels, int height, int width, int top_count, void* stream);
int gg_splat2d_f32(float* output, float* alpha_ws, const float* input, const float* coords,
                   const float* values, const float* sigma, int n, int num_points, int channels,
                   int height, int width, int soft_normalize, void* stream);
/* The reference's own C symbol, name and argument order unchanged (splat_gpu_impl.cuh:11-22: stream FIRST, void return),
 * so that the reference's torch binding utils/splat2d_cuda/src/splat_gpu.c:29-31 links against this library as it
 * stands.  `stream` is a hipStream_t.  An alias of gg_splat_forward_f32: same kernels, same results; a failure (which
 * the reference's void signature cannot report) is left in gg_last_error(). */
void SplatForwardGpu(void* stream, const float* bottom_coordinates, const float* bottom_values,
                     const float* bottom_sigma, float* top_alpha_splats, float* top_output, const int num_points_,
                     const int channels_, const int height_, const int width_, const int top_count);

/* ------------------------------------------------------------------------------------------
 * a6  anti-aliased sampling (MipmapWarp / Warp), models/spatial_transformers/antialiased_sampling.py.
 * padding_mode: 0 zeros, 1 border, 2 reflection (F.grid_sample, align_corners=False).
 * ------------------------------------------------------------------------------------------ */
/* One pyramid step: ReflectionPad2d(1) + depthwise [1,3,3,1]^2/64 stride 2 (:111-117).
 * in (planes,h,w) -> out (planes,h/2,w/2); h, w even. */
int gg_mip_downsample2x_f32(float* out, const float* in, int planes, int h, int w, void* stream);
/* Adjoint of the above: grad_in (planes,h,w) ACCUMULATES (+=) the back-projection of grad_out. */
int gg_mip_downsample2x_bwd_f32(float* grad_in, const float* grad_out, int planes, int h, int w, void* stream);

/* MipmapWarp.forward (:35-60) without materialising the (N,C*D,H,W) Gaussian stack: per output
 * pixel the mip level is computed from the grid (:62-97,181-210), the two bracketing levels are
 * sampled straight from the pyramid (bilinear upsample :155-160 folded into the tap fetch) and
 * blended (:212-238).
 *   pyr0        level 0: the (reflect-padded to a power of two, :130-137) input (N,C,hp,wp); pad_l the left/top pad
 *               (0 when h is 2^k)
 *   pyr_rest    levels 1 .. num_levels-1 consecutively in one buffer, level l as (N,C,hp>>l,wp>>l)   (ABI 5: the pyramid
 *               depth is an argument - ABI <= 4 took four fixed level pointers and refused max_level > 3, so the
 *               reference's DEFAULT constructor MipmapWarp(max_num_levels=8), antialiased_sampling.py:22, was not served)
 *   num_levels  levels the caller built: >= min(ceil(max_level) + 1, log2(hp) + 1), <= 8.  A pixel whose level lies
 *               beyond the pyramid's last (1 x 1) level - where the reference raises in ReflectionPad2d, :117 - samples it
 *   h, w        size of the ORIGINAL input (sampling coordinates refer to it)
 *   grid        (N,ho,wo,2) normalised coordinates
 *   out         (N,C,ho,wo);  levels_out (N,ho,wo) receives the clamped fractional level
 *   antialias   0 -> plain Warp (:9-16): level 0 everywhere (num_levels 1, pyr_rest NULL)
 *   max_level   = max_num_levels - 1 in [0, 7] (2.5 for the heads, warping_heads.py:32,170; 7 for the default)
 */
int gg_mipmap_warp_fwd_f32(float* out, float* levels_out, const float* pyr0, const float* pyr_rest, int num_levels,
                           const float* grid, int n, int c, int h, int w, int hp, int wp, int pad_l,
                           int ho, int wo, float max_level, float min_level,
                           int padding_mode, int antialias, void* stream);
/* The integer by-products of the sampling above, per output pixel (N,ho,wo) int32, from the same device functions
 * the forward / backward kernels use: ix_nw = floor(ix), iy_nw = floor(iy) with (ix, iy) the source coordinates after
 * grid_sampler_compute_source_index (ATen GridSampler.h:143-160: unnormalise, then clip / reflect for the padding
 * mode; 'zeros' leaves them unclipped, so the indices may lie outside the image), and floor / ceil of the clamped
 * mip level (antialiased_sampling.py:49,208-209,226-227).  Any output pointer may be NULL.  north_star's "bit-exact
 * warp grid indices" is tested on these (tests/test_gpu_indices.py).
 * level_arg (ABI 5): which neighbour (0 left, 1 right, 2 up, 3 down) holds the maximum coordinate distance the level
 * was computed from (:62-97: torch.max over the four distances, first maximum wins) - the point the level's
 * sub-gradient flows through; -1 without antialiasing. */
int gg_mipmap_warp_indices_f32(int* ix_nw, int* iy_nw, int* lvl_floor, int* lvl_ceil, int* level_arg,
                               const float* grid, int n, int h, int w, int ho, int wo, float max_level,
                               float min_level, int padding_mode, int antialias, void* stream);
/* Backward.  grad_grid (N,ho,wo,2) is overwritten; grad_pyr0 / grad_pyr_rest (the layout of pyr0 / pyr_rest)
 * ACCUMULATE (pass NULL for both to skip the image gradient).  Includes the gradient that reaches the grid through the
 * fractional mip level (levels % 1.0 is differentiable in the reference's autograd graph). */
int gg_mipmap_warp_bwd_f32(float* grad_grid, float* grad_pyr0, float* grad_pyr_rest, const float* grad_out,
                           const float* pyr0, const float* pyr_rest, int num_levels,
                           const float* grid, int n, int c, int h, int w, int hp, int wp, int pad_l,
                           int ho, int wo, float max_level, float min_level,
                           int padding_mode, int antialias, const signed char* level_arg_pin, void* stream);
/* level_arg_pin (ABI 5; NULL in product use): per output pixel (N,ho,wo) the neighbour 0..3 whose distance the level's
 * sub-gradient is routed through, or -1 for this evaluation's own arg-max.  Diagnostics: under a similarity warp the four
 * neighbour distances are exactly tied in real arithmetic and the arg-max is decided by the last ulp of the grid in every
 * implementation; with the reference's recorded decisions pinned the gradient is the reference's to rounding
 * (tests/test_gpu_stn_decisions.py). */

/* ------------------------------------------------------------------------------------------
 * a7  SimilarityHead.make_affine_matrix (warping_heads.py:36-56): params (N, 4*heads) = the regressed
 *     [rot | log-scale | shift_x | shift_y] blocks of `heads` columns each ->
 *     matrix (N, heads, 2, 3) = [[s cos r, -s sin r, tx], [s sin r, s cos r, ty]], r = pi * tanh(rot), s = exp(log-scale)
 *     (the reference: tanh, mul, exp, cos, sin, 4 products, neg, stack = 11 launches on (N, heads) tensors; ~25 in
 *     its backward).  bwd: grad_params (N, 4*heads) overwritten.
 * ------------------------------------------------------------------------------------------ */
int gg_similarity_matrix_f32(float* matrix, const float* params, int n, int heads, void* stream);
int gg_similarity_matrix_bwd_f32(float* grad_params, const float* grad_matrix, const float* params, int n, int heads,
                                 void* stream);

/* ------------------------------------------------------------------------------------------
 * a7  F.affine_grid(theta (N,2,3), (N,C,ho,wo), align_corners=False) (warping_heads.py:135,176)
 *     grid[n,i,j] = theta[n] . [x_j, y_i, 1],  x_j = linspace(-1,1,wo)[j]*(wo-1)/wo.
 * bwd: grad_theta (N,2,3) overwritten.
 * ------------------------------------------------------------------------------------------ */
int gg_affine_grid_f32(float* grid, const float* theta, int n, int ho, int wo, void* stream);
int gg_affine_grid_bwd_f32(float* grad_theta, const float* grad_grid, int n, int ho, int wo, void* stream);

/* ------------------------------------------------------------------------------------------
 * a8  flow composition of FlowHead.forward (warping_heads.py:180-193,239-243,268-277):
 *     delta = convex_upsample(low_flow, softmax_9(mask));  flow = identity + delta;
 *     flow  = [flow,1] @ base^T  (when base != NULL)
 *   low_flow (N,2,hl,wl)  - the flow_out conv output, channel-major (the reference permutes it to
 *                           (N,hl,wl,2) first, :198-199; reading NCHW directly removes that copy)
 *   mask     (N,9*ds*ds,hl,wl), ds = 8
 *   base     (N,2,3) or NULL
 *   delta, flow (N,ds*hl,ds*wl,2)
 * bwd: grad_low, grad_mask overwritten; grad_base (N,2,3) overwritten (NULL when base is NULL);
 *      g_flow / g_delta may each be NULL (treated as zero).
 * ------------------------------------------------------------------------------------------ */
int gg_flow_compose_fwd_f32(float* delta, float* flow, const float* low_flow, const float* mask,
                            const float* base, int n, int hl, int wl, int ds, void* stream);
int gg_flow_compose_bwd_f32(float* grad_low, float* grad_mask, float* grad_base,
                            const float* g_flow, const float* g_delta,
                            const float* low_flow, const float* mask, const float* base,
                            int n, int hl, int wl, int ds, void* stream);

/* Bilinear resize of a (N,hi,wi,2) flow field by `scale` = ho/hi (F.interpolate(scale_factor=...,
 * mode='bilinear', align_corners=False) on the permuted flow, warping_heads.py:250).  bwd accumulates
 * nothing: grad_in is overwritten. */
int gg_flow_resize_f32(float* out, const float* in, int n, int hi, int wi, int ho, int wo, float scale, void* stream);
int gg_flow_resize_bwd_f32(float* grad_in, const float* grad_out, int n, int hi, int wi, int ho, int wo, float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * a9  BilinearDownsample(stride).forward (antialiased_sampling.py:241-256): reflect-pad stride/2,
 *     separable tent [1,3,..,3,1]/sum, decimate by stride.  in (planes,h,w) -> out (planes,h/stride,w/stride).
 * ------------------------------------------------------------------------------------------ */
int gg_bilinear_downsample_f32(float* out, const float* in, int planes, int h, int w, int stride, void* stream);
int gg_bilinear_downsample_bwd_f32(float* grad_in, const float* grad_out, int planes, int h, int w, int stride, void* stream);

/* ------------------------------------------------------------------------------------------
 * a10  flow regularisers (models/losses/loss.py:4-18) in one pass over delta (N,hf,wf,2):
 *      losses[0] = total_variation_loss (Huber(1) of forward differences, mean over each difference
 *      tensor, x + y), losses[1] = flow_identity_loss (mean square).  losses (2,) overwritten.
 * bwd: grad_delta = g_tv * dTV/ddelta + g_id * dID/ddelta  (g_* are the upstream scalars, on device).
 * ------------------------------------------------------------------------------------------ */
int gg_flow_losses_f32(float* losses, const float* delta, int n, int hf, int wf, void* stream);
int gg_flow_losses_bwd_f32(float* grad_delta, const float* delta, const float* g_losses, int n, int hf, int wf, void* stream);

/* ------------------------------------------------------------------------------------------
 * a3/a4  convolutions as implicit GEMM on the fp32 MFMA pipe (v_mfma_f32_32x32x2_f32).
 * Replaces conv2d_gradfix.conv2d / conv_transpose2d (models/stylegan2/op/conv2d_gradfix.py:22-75)
 * and, through in_scale / out_scale, the modulate / demodulate arithmetic of
 * ModulatedConv2d.forward (models/stylegan2/networks.py:233-282) in its shared-weight form:
 *     y[n,co] = out_scale[n,co] * sum_{ci,ky,kx} W[co,ci,ky,kx] * (in_scale[n,ci] * x[n,ci,..])
 *
 * Weights are consumed in GEMM layout wmat (groups, K, Cout) with K = Cin_g*kh*kw ordered
 * (ci, ky, kx), produced by gg_conv_pack_weight_f32 from the torch layout.
 *   x        (batch, groups*cin_g, h, w)
 *   y        (batch, groups*cout_g, ho, wo)
 *   in_scale (batch, groups*cin_g) or NULL;  out_scale (batch, groups*cout_g) or NULL
 *   bias     (groups*cout_g) or NULL, added after out_scale
 *   mode 0: correlation  ho = (h + 2*pad - k)/stride + 1
 *   mode 1: transposed   ho = (h-1)*stride - 2*pad + k      (stride 2; stride-1 transposed convolution is
 *           mode 0 with flipped taps).  out_h / out_w: 0 = natural size; for mode 1 a value up to
 *           natural + stride - 1 plays the role of output_padding (extra rows / columns are zero).
 *   ksize in {1, 3}.
 * ------------------------------------------------------------------------------------------ */
int gg_conv_pack_weight_f32(float* wmat, const float* w, int groups, int cout_g, int cin_g, int kh, int kw,
                            int transpose_io, int flip, float scale, void* stream);
int gg_conv2d_f32(float* y, const float* x, const float* wmat, const float* in_scale, const float* out_scale,
                  const float* bias, int batch, int groups, int cin_g, int cout_g, int h, int w,
                  int ksize, int stride, int pad, int mode, int out_h, int out_w, void* stream);
/* Split-precision variant of gg_conv2d_f32 on the bf16 matrix pipe (16x the fp32 MFMA rate): every fp32
 * operand is split into `limbs` bf16 limbs and the product is assembled from the limb pairs (i,j), i+j <
 * limbs, accumulated in fp32.  limbs = 2: 3 MFMAs, error ~2^-16 per product; limbs = 3: 6 MFMAs, fp32-class.
 * limbs = 18 (= 16 + 2; ABI 2): two BINARY16 limbs (11 + 11 significand bits, v_mfma_f32_32x32x16_f16), 3 MFMAs,
 * error ~2^-22 per product.  For FORWARD convolutions only: binary16 has no exponent range to spare, so the pack
 * multiplies the weights by 2^8 (the kernels' epilogues divide the fp32 accumulator by 2^8 - exact) and the leading
 * limb of an operand is rounded toward zero, so it saturates at +-65504 instead of overflowing.  Operand range
 * (activation times in_scale): full precision for |x| <= 65504; 2^-12 up to ~1.3e5; beyond that the LOW limb
 * overflows and the output is inf / NaN (loud, never silently wrong); activations below ~1e-7 vanish.  Gradient
 * tensors span more than that: pass limbs = 2 for them (gg_conv3x3_masked_dgrad_f32 and the weight-gradient entry
 * points do not take 18).  A weight pack made with limbs = 18 must be used with limbs = 18 and vice versa.
 * (Round 4: every tile carries a per-chunk block exponent taken from the data, so the range statement above is history
 * - any finite operand is staged at full limb precision - and data gradients run on code 18 too.)
 * limbs = 50 (= 32 + 18; ABI 5): as 18, and the operand x is a GRADIENT: the block exponent leaves a chunk unscaled
 * (E = 0) only for amax in [2^5, 2^11] instead of [2^-3, 2^11] - unscaled, an entry keeps an absolute 2^-25, which an
 * activation bound never sees but which is 3e-3 of a 1e-5 gradient entry next to an outlier of 0.5
 * (tests/test_gpu_block_exponent_band.py).  The masked data-gradient entry points imply it.  gg_conv_pack_weight_split
 * takes 18 for the pack of either.
 * Weights come pre-split from gg_conv_pack_weight_split: bf16 planes wsplit[limb][g][co][k], K ordered
 * (tap, ci) with ci fastest, limb planes `limb_stride` elements apart (= groups*cout_g*cin_g*kh*kw).
 * Requires cin_g % 32 == 0; every other argument as gg_conv2d_f32. */
int gg_conv_pack_weight_split(unsigned short* wsplit, const float* w, int groups, int cout_g, int cin_g, int kh,
                              int kw, int transpose_io, int flip, float scale, int limbs, void* stream);
/* Data gradient of a "3x3 conv + leaky ReLU" layer with the activation's backward fused into the gather:
 *   y = out_scale * conv3x3(W, in_scale * x * (mask_ref > 0 ? 1 : alpha) * gain)      (stride 1, pad 1, split precision)
 * x = the incoming gradient, mask_ref = the layer's saved activation OUTPUT (same shape as x), W = the data-gradient
 * pack (flipped / transposed taps).  Only the patch-reuse kernel carries the mask: the call returns GG_NOT_SERVED and
 * launches NOTHING when the shape is not served by it (caller: gg_fused_lrelu_bwd_f32 followed by gg_conv2d_split_f32). */
int gg_conv3x3_masked_dgrad_f32(float* y, const float* x, const float* mask_ref, float alpha, float gain,
                                const unsigned short* wsplit, long long limb_stride, int limbs, const float* in_scale,
                                const float* out_scale, int batch, int cin, int cout, int h, int w, void* stream);
/* Many weight packs in one launch.  `jobs`: device array of `njobs` records
 *   struct { void* dst; const float* src; long long total, limb_stride;
 *            int cout_g, cin_g, kh, kw, transpose_io, flip, limbs; float scale; }          (64 bytes each)
 * limbs = 0 writes the fp32 GEMM layout of gg_conv_pack_weight_f32, 2 | 3 the bf16 limb planes of
 * gg_conv_pack_weight_split (total = groups*cout_g*cin_g*kh*kw, limb_stride = elements between limb planes). */
int gg_conv_pack_weights_many(const void* jobs, int njobs, void* stream);
int gg_conv2d_split_f32(float* y, const float* x, const unsigned short* wsplit, long long limb_stride, int limbs,
                        const float* in_scale, const float* out_scale, const float* bias, int batch, int groups,
                        int cin_g, int cout_g, int h, int w, int ksize, int stride, int pad, int mode, int out_h,
                        int out_w, void* stream);
/* Round 6 (second half), ResBlock (networks.py:375-393) without its two element-wise passes:
 *   gg_conv2d_split_act_f32: gg_conv2d_split_f32 followed by lrelu(y + act_bias[c], alpha) * gain inside the library - in
 *     the stride-2 patch tile's epilogue, in the split-K reduce pass, or (generic tile without split-K) as the library's
 *     own in-place pass.  H_out * W_out must be a multiple of 4 (GG_NOT_SERVED otherwise).
 *   gg_conv1x1_split_residual_f32: y = conv1x1(x) (* out_scale, + bias) + residual, residual (N, Cout, H_out, W_out) - the
 *     skip branch's 1x1 convolution with the residual merge (out + skip) in its epilogue / reduce pass. */
int gg_conv2d_split_act_f32(float* y, const float* x, const unsigned short* wsplit, long long limb_stride, int limbs,
                            const float* in_scale, const float* out_scale, const float* act_bias, float alpha, float gain,
                            int batch, int groups, int cin_g, int cout_g, int h, int w, int ksize, int stride, int pad,
                            int mode, int out_h, int out_w, void* stream);
int gg_conv1x1_split_residual_f32(float* y, const float* x, const unsigned short* wsplit, long long limb_stride, int limbs,
                                  const float* in_scale, const float* out_scale, const float* bias, const float* residual,
                                  int batch, int groups, int cin_g, int cout_g, int h, int w, int stride, void* stream);
/* StyledConv without upsampling in ONE pass (networks.py:243-298, 344-350: ModulatedConv2d -> NoiseInjection ->
 * FusedLeakyReLU):  y = lrelu(out_scale[n,co] * conv3x3(W, in_scale[n,ci] * x) + noise_weight[0] * noise[n,0] +
 * act_bias[co], alpha) * gain.  3x3 / stride 1 / pad 1, one group.  limbs = 0: fp32 MFMA kernel with `wmat`
 * (gg_conv_pack_weight_f32); limbs = 2|3: split-precision kernel with `wsplit` (gg_conv_pack_weight_split).
 * The activation rides in the convolution's epilogue when the launch needs no split-K; otherwise in the pass that
 * adds the split-K partial sums - the result is the same either way.  H*W % 4 == 0.
 * noise = NULL drops the noise term and in_scale / out_scale / act_bias may be NULL too, which makes this the plain
 * "3x3 convolution + bias + leaky ReLU" of the STN trunk (EqualConv2d + FusedLeakyReLU, networks.py:602-640) and,
 * with alpha = 0 and gain = 1, the conv + bias + ReLU of the VGG16 backbone (lpips_backbones.py:98-140). */
int gg_modconv3x3_act_f32(float* y, const float* x, const float* wmat, const unsigned short* wsplit,
                          long long limb_stride, int limbs, const float* in_scale, const float* out_scale,
                          const float* noise, const float* noise_weight, const float* act_bias, float alpha,
                          float gain, int batch, int cin, int cout, int h, int w, void* stream);
/* ABI 4: the same launch, additionally writing the SIGN PLANE of its output: one bit per element,
 *   bit (co & 31) of sign_bits[(n * H * W + y * W + x) * (cout / 32) + co / 32] = (y[n, co, y, x] > 0),
 * i.e. exactly what the leaky-ReLU backward tests (op/fused_act.py:33-38 -> fused_bias_act_kernel.cu:36-47 take the
 * saved OUTPUT only as a sign).  gg_conv3x3_masked_dgrad_bits_f32 consumes it: the data gradient of the layer then reads
 * one word per (pixel, 32 channels) where gg_conv3x3_masked_dgrad_f32 reads 32 floats.  The plane comes out of the 3x3
 * patch tile's own activation epilogue: gg_last_sign_bits_written() (calling thread's last launch) is 1 when the launch
 * produced it, 0 when another kernel served the shape (split-K, the fp32 kernel, cout % 32 != 0) - the caller then keeps
 * mask_ref.  sign_bits: N * H * W * (cout / 32) words, or NULL. */
int gg_modconv3x3_act_bits_f32(float* y, const float* x, const float* wmat, const unsigned short* wsplit,
                               long long limb_stride, int limbs, const float* in_scale, const float* out_scale,
                               const float* noise, const float* noise_weight, const float* act_bias, float alpha,
                               float gain, int batch, int cin, int cout, int h, int w, unsigned int* sign_bits,
                               void* stream);
int gg_last_sign_bits_written(void);

/* ------------------------------------------------------------------------------------------
 * ABI 5 (round 6): the up-sampling convolution's operand written ONCE in MFMA-ready form.
 * Every consumer tile of the split-precision kernels re-does, per 16 / 32-channel chunk of its patch, what turns an fp32
 * activation into matrix-pipe operands: multiply by the layer's style, reduce the largest magnitude (block exponent),
 * split into two binary16 limbs, transpose NCHW -> channel-fastest LDS rows.  For the generator's transposed 3x3 /
 * stride-2 convolutions (networks.py:254-266) that work is moved into the pass that ALREADY streams the same activation
 * for the ToRGB layer (networks.py:352-372):
 *   gg_modconv3x3_act_amax_f32   the fused StyledConv launch of gg_modconv3x3_act_bits_f32, additionally leaving
 *                                amax_out[n] = max |y[n]| (atomic max of non-negative bit patterns: order-independent;
 *                                the caller zeroes amax_out; gg_last_amax_written() = 1 when the 3x3 patch tile without
 *                                split-K served the launch and wrote it - else the caller keeps the fp32 route)
 *   gg_torgb_limb_f32            rgb = ToRGB(y) (modulated 1x1 to 3 channels + bias) AND y's limb form for the next
 *                                layer: xlimb[n][cin/16][pixel][limb 0|1][16 channels] binary16 = split(y * next_style[n]
 *                                * 2^-E[n]), xlimb_exp[n] = E[n] = the block-exponent rule of the consumer applied to
 *                                amax[n] * max_c |next_style[n][c]|.  cin % 16 == 0, cin <= 1024, hw % 4 == 0, else
 *                                GG_NOT_SERVED
 *   gg_convT3x3s2_prelimb_f32    conv_transpose2d(3x3, stride 2) of that operand with the binary16 pack `wsplit` (code
 *                                18), epilogue scale 2^E[n] * out_scale + bias; served by the 16-channel-chunk tile only
 *                                (power-of-two w >= 8, cin % 32 == 0): GG_NOT_SERVED, nothing launched, otherwise.
 * With E = 0 (activations within [2^-3, 2^11]) the result is BITWISE the fp32-operand kernel's; measured on the consumer:
 * 1.16 - 1.27x, VALU instructions per MFMA 3.18 -> 1.48 (profiles/r06_c_prelimb_probe.txt).
 * ------------------------------------------------------------------------------------------ */
int gg_modconv3x3_act_amax_f32(float* y, const float* x, const float* wmat, const unsigned short* wsplit,
                               long long limb_stride, int limbs, const float* in_scale, const float* out_scale,
                               const float* noise, const float* noise_weight, const float* act_bias, float alpha,
                               float gain, int batch, int cin, int cout, int h, int w, unsigned int* sign_bits,
                               float* amax_out, void* stream);
int gg_last_amax_written(void);
int gg_torgb_limb_f32(float* rgb, unsigned short* xlimb, int* xlimb_exp, const float* y, const float* rgb_wmat,
                      const float* rgb_style, const float* rgb_bias, const float* next_style, const float* amax,
                      int batch, int cin, long long hw, void* stream);
int gg_convT3x3s2_prelimb_f32(float* y, const unsigned short* xlimb, const int* xlimb_exp,
                              const unsigned short* wsplit, long long limb_stride, const float* out_scale,
                              const float* bias, int batch, int cin, int cout, int h, int w, int pad, int out_h,
                              int out_w, void* stream);
/* gg_conv3x3_masked_dgrad_f32 with the mask taken from the sign plane of gg_modconv3x3_act_bits_f32 (words per pixel =
 * cin / 32; cin = the reduction channels of this launch = the layer's output channels).  limbs = 18 only (the
 * binary16-limb patch tiles); GG_NOT_SERVED otherwise and when the patch tile does not cover the shape.  Results are
 * bitwise equal to gg_conv3x3_masked_dgrad_f32 on the fp32 output the plane was taken from. */
int gg_conv3x3_masked_dgrad_bits_f32(float* y, const float* x, const unsigned int* mask_bits, float alpha, float gain,
                                     const unsigned short* wsplit, long long limb_stride, int limbs,
                                     const float* in_scale, const float* out_scale, int batch, int cin, int cout, int h,
                                     int w, void* stream);
/* Round 6 (second half): the perceptual trunk's RGB stem (lpips_backbones.py:109: Conv2d(3, 64, 3, padding=1) + ReLU).
 * Forward: gg_modconv3x3_act_bits_f32 with limbs = 0 and cin <= 4 runs a streaming few-input-channel kernel that applies
 * bias + (leaky) ReLU in registers and writes the sign plane (gg_last_sign_bits_written() == 1).
 * Backward: dx (N, cout <= 4, H, W) = conv3x3(dy * lrelu'(y), wmat) with lrelu'(y) read from that plane (words per pixel =
 * cin / 32; cin = this launch's reduction channels = the layer's output channels, a multiple of 32) - instead of
 * threshold_backward / gg_fused_lrelu_bwd_f32 over the 134 MB gradient followed by the few-output-channel convolution.
 * wmat: the fp32 GEMM layout of the data-gradient convolution (gg_conv_pack_weight_f32, transpose_io = flip = 1).
 * GG_NOT_SERVED (nothing launched) for shapes the few-output-channel kernel does not take. */
int gg_conv3x3_fewout_masked_bits_f32(float* dx, const float* dy, const unsigned int* mask_bits, float alpha, float gain,
                                      const float* wmat, int batch, int cin, int cout, int h, int w, void* stream);
/* Weight gradient: dw (groups, cout_g, cin_g, k, k) torch layout, overwritten.  (All weight-gradient entry points:
 * the K-splits' partial tiles go to the library's scratch and are added in split order.)
 *   dw[g,co,ci,ky,kx] = sum_{n,oy,ox} dy[n,g*cout_g+co,oy,ox] * x[n,g*cin_g+ci, oy*stride+ky-pad, ox*stride+kx-pad] */
int gg_conv2d_wgrad_f32(float* dw, const float* x, const float* dy, int batch, int groups, int cin_g, int cout_g,
                        int h, int w, int ksize, int stride, int pad, float scale, void* stream);
/* Split-precision weight gradient (bf16 matrix pipe, `limbs` bf16 limbs per operand; see gg_conv2d_split_f32).
 * Requires OH*OW % 32 == 0, OW % 4 == 0 and a 16-byte aligned dy. */
int gg_conv2d_wgrad_split_f32(float* dw, const float* x, const float* dy, int batch, int groups, int cin_g,
                              int cout_g, int h, int w, int ksize, int stride, int pad, float scale, int limbs,
                              void* stream);
/* dw += weight gradient (limbs = 0: fp32 MFMA kernel; 2|3: split precision, same preconditions as above).  The
 * optimizer-facing form: the trainer keeps all gradients in one flat zero-initialised arena and every layer adds
 * its weight gradient straight into its slice (no per-layer memset, no separate accumulation pass). */
int gg_conv2d_wgrad_acc_f32(float* dw, const float* x, const float* dy, int batch, int groups, int cin_g,
                            int cout_g, int h, int w, int ksize, int stride, int pad, float scale, int limbs,
                            void* stream);
/* Weight gradient, kernel chosen by shape; K-split partial tiles are summed in a fixed order (no float atomics).
 * With limbs != 0, a 3x3 / stride-1 / pad-1 convolution and W % 32 == 0 (or W == 16) the row-streaming kernel runs:
 * each block walks down a 32-pixel-wide strip keeping a rolling 3-row window of x in LDS and writes its partial
 * (co, ci, tap) tile to the workspace, a second kernel sums the partials (dw = or += depending on `accumulate`); 1x1
 * convolutions with <= 4 input channels use a streaming reduction; any other shape is gg_conv2d_wgrad(_split / _acc)_f32.
 * `workspace` may be NULL (or too small): the library's per-stream scratch is used instead.  Caller-provided need:
 * blocks * 147,456 bytes (at most ~1,100 blocks). */
int gg_conv2d_wgrad_ws_f32(float* dw, const float* x, const float* dy, int batch, int groups, int cin_g, int cout_g,
                           int h, int w, int ksize, int stride, int pad, float scale, int limbs, int accumulate,
                           float* workspace, long long workspace_bytes, void* stream);
/* Weight (and bias) gradient of a "3x3 conv + bias + leaky ReLU" layer from the gradient of its OUTPUT: the
 * activation's backward dy * (mask_ref > 0 ? 1 : alpha) * gain is applied while dy is staged (mask_ref = the saved
 * output), dw (=/+=) as gg_conv2d_wgrad_ws_f32 and, if dbias != NULL, dbias[co] += sum of the masked gradient
 * (dbias must be initialised by the caller).  Returns GG_NOT_SERVED and launches nothing when the row-streaming kernel does not
 * serve the shape (W % 32 != 0, limbs != 2, workspace too small ...). */
int gg_conv3x3_masked_wgrad_f32(float* dw, float* dbias, const float* x, const float* dy, const float* mask_ref,
                                float alpha, float gain, int batch, int cin, int cout, int h, int w, float scale,
                                int limbs, int accumulate, float* workspace, long long workspace_bytes, void* stream);
/* Style modulation of one ModulatedConv2d layer in one launch (networks.py:214-216 EqualLinear + :244-249):
 *   style[n,ci] = sum_k latent[n*lat_stride + k] * w[ci,k] * w_scale + b[ci] * b_scale      (b may be NULL)
 *   demod[n,co] = rsqrt(sum_ci style[n,ci]^2 * wsq[co,ci] + eps)                             (demod may be NULL)
 * wsq (cout, cin) = sum over taps of (conv weight * scale)^2.  style_dim, cin <= 2048. */
int gg_style_demod_f32(float* style, float* demod, const float* latent, long long lat_stride, const float* w,
                       const float* b, const float* wsq, int n, int style_dim, int cin, int cout, float w_scale,
                       float b_scale, float eps, void* stream);
/* The modulation vectors of many frozen layers in two launches (networks.py:214-216,244-249 for every layer of a
 * generator pass).  style_jobs / demod_jobs: device arrays of 56-byte records {const float* m; const float* bias;
 * int64 out_off; int64 in_off; int32 kdim, rows; float scale, bias_scale, eps; int32 pad}: outputs go to
 * out_base + out_off as (n, rows); a style job reads latent + in_off * slot_stride (sample stride
 * lat_sample_stride), a demodulation job reads the style at out_base + in_off.  kdim <= 512. */
int gg_style_bank_f32(float* out_base, const float* latent, long long lat_sample_stride, int slot_stride,
                      const void* style_jobs, int n_style_jobs, int max_style_rows, const void* demod_jobs,
                      int n_demod_jobs, int max_demod_rows, int n, void* stream);

/* Perceptual-loss tail of one feature tap (reference models/losses/lpips.py:26-28, 190-199):
 * feats (2n, c, hw): samples [0,n) belong to image 0, [n,2n) to image 1.  With u = f / (sqrt(sum_c f^2) + eps):
 *   out[s] = mean_pixels sum_c lin[c] * (u0 - u1)^2     (lin NULL = all ones: the lpips=False / vgg_ssl branch)
 * _bwd writes d out / d feats (2n, c, hw) for grad_out (n). */
int gg_lpips_tail_fwd_f32(float* out, const float* feats, const float* lin, int n, int c, long long hw, float eps,
                          void* stream);
int gg_lpips_tail_bwd_f32(float* dfeats, const float* feats, const float* lin, const float* grad_out, int n, int c,
                          long long hw, float eps, int accumulate, void* stream);

/* 2x2 / stride-2 max pooling of the VGG16 perceptual-loss trunk (models/losses/lpips_backbones.py:101-121: torchvision's
 * `features` -> ATen max_pool2d: row-major window scan, a later element wins only if strictly greater or NaN).
 * x: (planes, h, w) dense, h and w even.  `code` (planes * h/2 * w/2 bytes) receives the winner's position 0..3 and
 * is all the backward needs: dx (planes, h, w) is written in full from dy and code (no zero fill, no atomics). */
int gg_maxpool2x2_fwd_f32(float* out, unsigned char* code, const float* x, long long planes, int h, int w,
                          void* stream);
int gg_maxpool2x2_bwd_f32(float* dx, const float* dy, const unsigned char* code, long long planes, int h, int w,
                          void* stream);
/* Data gradient of a modulated 1x1 ToRGB convolution (networks.py:352-372, no demodulation) ADDED into an existing
 * gradient:  g[n,c,p] += sum_{k<3} w[k,c] * wscale * style[n,c] * grad_rgb[n,k,p].
 * g (n,c,hw) accumulates, grad_rgb (n,3,hw), w (3,c) = the ToRGB weight, style (n,c) its modulation; hw % 4 == 0,
 * n*c <= 65535.  Used where the activation feeds both the next layer and ToRGB: the next layer's data gradient is
 * the running sum and the ToRGB branch is added in one read-modify-write pass. */
int gg_torgb_dgrad_add_f32(float* g, const float* grad_rgb, const float* w, const float* style, float wscale, int n,
                           int c, long long hw, void* stream);
/* Style gradient of the shared-weight modulated convolution from its per-plane dot products, one launch:
 *   dstyle[n,ci] = dot_x[n,ci] + 2 style[n,ci] * sum_co (-0.5 dot_y[n,co] demod[n,co]^2) wsq[co,ci]
 * dot_x = <d(style-scaled input), input> (n, cin), dot_y = <d output, output> (n, cout); cout <= 1024.
 * (the autograd of networks.py:243-249 w.r.t. `style`, which the reference obtains through the per-sample weights) */
int gg_modconv_style_grad_f32(float* dstyle, const float* dot_x, const float* dot_y, const float* demod,
                              const float* style, const float* wsq, int n, int cin, int cout, void* stream);
/* Per-(n,c) dot products over the spatial plane: out[n*c] = sum_hw a*b (style / demod gradients). */
int gg_plane_dot_f32(float* out, const float* a, const float* b, int planes, long long hw, void* stream);

/* ------------------------------------------------------------------------------------------
 * a13  optimiser step over flat parameter arenas: Adam (torch.optim.Adam semantics, train.py:204-205)
 *      fused with the EMA update accumulate(t_ema, t, decay) (models/__init__.py:19-24, train.py:134).
 *      ema may be NULL (Adam only).  step is the 1-based step count (bias correction).
 * ------------------------------------------------------------------------------------------ */
int gg_adam_ema_f32(float* param, float* exp_avg, float* exp_avg_sq, float* ema, const float* grad,
                    long long numel, float lr, float beta1, float beta2, float eps, int step,
                    float ema_decay, float grad_scale, void* stream);
/* The same update with the step-dependent scalars in DEVICE memory: hyper = {lr, 1 - beta1^t, sqrt(1 - beta2^t),
 * grad_scale}.  Lets a captured hipGraph of the whole training step be replayed while the host only refreshes four
 * floats per step (learning-rate schedule of train.py:129-132, Adam's bias corrections). */
int gg_adam_ema_dev_f32(float* param, float* exp_avg, float* exp_avg_sq, float* ema, const float* grad,
                        long long numel, const float* hyper, float beta1, float beta2, float eps, float ema_decay,
                        void* stream);


#ifdef __cplusplus
}
#endif
#endif /* GANGEALING_HIP_H */
