// a11  splat2d: Gaussian soft-scatter of (x, y, value) points (forward only, as in the reference).
//
// Semantics follow utils/splat2d_cuda/src/splat_gpu_impl.cu:41-96 and splat_gpu.c:12-42:
//   points outside [0,W) x [0,H) are dropped; the footprint is the +-2*sigma box clipped to the
//   image; weight = exp(-((px-x)^2 + (py-y)^2) / (2 sigma^2)); alpha and alpha*value are
//   accumulated with atomics; the host glue then divides by (alpha [clamped >= 1] + 1e-8).
//
// MI355X mapping: the reference runs one THREAD per point (32-thread blocks, a serial loop over
// the ~7x7 footprint).  Here a 16-lane quarter of a 64-lane wave owns a point and its lanes fan out over
// the footprint pixels, so a footprint row becomes a run of consecutive addresses for the L2
// atomic units (global_atomic_add_f32 via unsafeAtomicAdd - no CAS loop).  Sixteen lanes suit both
// ends of the range the applications use: sigma = 0.3 touches <= 3x3 pixels (one pass, 9 of 16 lanes;
// a whole wave per point would idle 55 of 64), sigma = 1.3 touches 7x7 (four passes, 49 of 64 lane slots).
// The accumulation order of overlapping points is the arrival order of the atomics: results vary in
// the last bits from run to run, exactly as the reference kernel's do (inference-side operator; not on
// the training path).
#include "../../include/gangealing_hip.h"
#include "gg_common.h"

namespace {

__global__ __launch_bounds__(256) void splat_forward_kernel(
    const float* __restrict__ coords, const float* __restrict__ values, const float* __restrict__ sigma,
    float* __restrict__ alpha_splats, float* __restrict__ output, int num_points, int channels, int height,
    int width, int top_count) {
  constexpr int G = 16;                                   // lanes per point
  const int sub = threadIdx.x & (G - 1);
  const long long group = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const long long ngroups = ((long long)gridDim.x * blockDim.x) / G;
  const size_t hw = (size_t)height * width;
  for (long long index = group; index < top_count; index += ngroups) {
    const int n = (int)(index / num_points);
    const float xc = coords[2 * (size_t)index];
    const float yc = coords[2 * (size_t)index + 1];
    const float stdev = sigma[n];
    if (!(xc >= 0.f && xc < (float)width && yc >= 0.f && yc < (float)height)) continue;
    const float length = 2.f * stdev;
    const float normalizer = -(1.f / (2.f * stdev * stdev));
    const int t = (int)fmaxf(0.f, floorf(yc - length));
    const int b = (int)fminf((float)(height - 1), ceilf(yc + length));
    const int l = (int)fmaxf(0.f, floorf(xc - length));
    const int r = (int)fminf((float)(width - 1), ceilf(xc + length));
    const int bw = r - l + 1, bh = b - t + 1;
    if (bw <= 0 || bh <= 0) continue;
    const int area = bw * bh;
    const float* val = values + (size_t)index * channels;
    float* a_img = alpha_splats + (size_t)n * hw;
    float* o_img = output + (size_t)n * channels * hw;
    for (int p = sub; p < area; p += G) {
      const int dy = p / bw, dx = p - dy * bw;
      const int lh = t + dy, lw = l + dx;
      const float fx = (float)lw - xc, fy = (float)lh - yc;
      const float alpha = expf(normalizer * (fx * fx + fy * fy));
      const size_t pix = (size_t)lh * width + lw;
      unsafeAtomicAdd(a_img + pix, alpha);
      for (int c = 0; c < channels; ++c) unsafeAtomicAdd(o_img + (size_t)c * hw + pix, alpha * val[c]);
    }
  }
}

__global__ __launch_bounds__(256) void splat_normalize_kernel(float* __restrict__ output,
                                                              const float* __restrict__ alpha, long long total,
                                                              int channels, long long hw, int soft) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long pix = i % hw;
    const long long n = i / (hw * channels);
    float a = alpha[n * hw + pix];
    if (soft) a = fmaxf(a, 1.0f);
    output[i] = output[i] / (a + 1e-8f);
  }
}

}  // namespace

extern "C" int gg_splat_forward_f32(const float* coords, const float* values, const float* sigma,
                                    float* alpha_splats, float* output, int num_points, int channels, int height,
                                    int width, int top_count, void* stream) {
  if (top_count <= 0) return 0;
  if (!coords || !values || !sigma || !alpha_splats || !output || num_points <= 0)
    return gg::fail(-2, "splat_forward: bad arguments");
  const long long threads = (long long)top_count * 16;
  splat_forward_kernel<<<gg::stream_grid(threads, 256), 256, 0, gg::as_stream(stream)>>>(
      coords, values, sigma, alpha_splats, output, num_points, channels, height, width, top_count);
  return gg::launch_status("splat_forward");
}

extern "C" int gg_splat2d_f32(float* output, float* alpha_ws, const float* input, const float* coords,
                              const float* values, const float* sigma, int n, int num_points, int channels,
                              int height, int width, int soft_normalize, void* stream) {
  const long long hw = (long long)height * width;
  const long long total = (long long)n * channels * hw;
  if (total <= 0) return 0;
  if (!output || !alpha_ws || !input) return gg::fail(-2, "splat2d: null pointer");
  hipStream_t st = gg::as_stream(stream);
  hipError_t e = hipMemcpyAsync(output, input, sizeof(float) * (size_t)total, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return gg::fail((int)e, "splat2d: copy failed: %s", hipGetErrorString(e));
  e = hipMemsetAsync(alpha_ws, 0, sizeof(float) * (size_t)n * hw, st);
  if (e != hipSuccess) return gg::fail((int)e, "splat2d: memset failed: %s", hipGetErrorString(e));
  int rc = gg_splat_forward_f32(coords, values, sigma, alpha_ws, output, num_points, channels, height, width,
                                n * num_points, stream);
  if (rc) return rc;
  splat_normalize_kernel<<<gg::stream_grid(total, 256), 256, 0, st>>>(output, alpha_ws, total, channels, hw,
                                                                      soft_normalize);
  return gg::launch_status("splat_normalize");
}
