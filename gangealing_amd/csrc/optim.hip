// a13  Adam + EMA over flat fp32 arenas: one streaming pass (reads p, g, m, v, ema; writes p, m, v,
// ema = 36 B/element) instead of the reference's per-tensor Python loops (130 EMA launches +
// per-tensor Adam, models/__init__.py:19-24, train.py:126-134).
#include "../../include/gangealing_hip.h"
#include "gg_common.h"

namespace {

__global__ __launch_bounds__(256) void adam_ema_kernel(float* __restrict__ p, float* __restrict__ m,
                                                       float* __restrict__ v, float* __restrict__ ema,
                                                       const float* __restrict__ g, long long n, float lr,
                                                       float beta1, float beta2, float eps, float bc1,
                                                       float bc2_sqrt, float decay, float gscale) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float grad = g[i] * gscale;
    const float mi = beta1 * m[i] + (1.f - beta1) * grad;            // exp_avg.lerp_(grad, 1-beta1)
    const float vi = beta2 * v[i] + (1.f - beta2) * grad * grad;     // exp_avg_sq.mul_(b2).addcmul_(g,g,1-b2)
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    const float pi = p[i] - (lr / bc1) * (mi / denom);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi;
    if (ema) ema[i] = ema[i] * decay + pi * (1.f - decay);
  }
}

// The same update with the step-dependent scalars read from device memory: hyper = {lr, 1 - beta1^t,
// sqrt(1 - beta2^t), grad_scale}.  A captured hipGraph of the training step replays this launch unchanged while the
// host refreshes the four floats before every replay (learning-rate schedule, bias corrections).
__global__ __launch_bounds__(256) void adam_ema_dev_kernel(float* __restrict__ p, float* __restrict__ m,
                                                           float* __restrict__ v, float* __restrict__ ema,
                                                           const float* __restrict__ g, long long n,
                                                           const float* __restrict__ hyper, float beta1, float beta2,
                                                           float eps, float decay) {
  const float lr = hyper[0], bc1 = hyper[1], bc2_sqrt = hyper[2], gscale = hyper[3];
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float grad = g[i] * gscale;
    const float mi = beta1 * m[i] + (1.f - beta1) * grad;
    const float vi = beta2 * v[i] + (1.f - beta2) * grad * grad;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    const float pi = p[i] - (lr / bc1) * (mi / denom);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi;
    if (ema) ema[i] = ema[i] * decay + pi * (1.f - decay);
  }
}

}  // namespace

extern "C" int gg_adam_ema_dev_f32(float* param, float* exp_avg, float* exp_avg_sq, float* ema, const float* grad,
                                   long long numel, const float* hyper, float beta1, float beta2, float eps,
                                   float ema_decay, void* stream) {
  if (numel <= 0) return 0;
  if (!param || !exp_avg || !exp_avg_sq || !grad || !hyper) return gg::fail(-2, "adam_ema_dev: bad arguments");
  adam_ema_dev_kernel<<<gg::stream_grid(numel, 256), 256, 0, gg::as_stream(stream)>>>(
      param, exp_avg, exp_avg_sq, ema, grad, numel, hyper, beta1, beta2, eps, ema_decay);
  return gg::launch_status("adam_ema_dev");
}

extern "C" int gg_adam_ema_f32(float* param, float* exp_avg, float* exp_avg_sq, float* ema, const float* grad,
                               long long numel, float lr, float beta1, float beta2, float eps, int step,
                               float ema_decay, float grad_scale, void* stream) {
  if (numel <= 0) return 0;
  if (!param || !exp_avg || !exp_avg_sq || !grad || step < 1) return gg::fail(-2, "adam_ema: bad arguments");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  adam_ema_kernel<<<gg::stream_grid(numel, 256), 256, 0, gg::as_stream(stream)>>>(
      param, exp_avg, exp_avg_sq, ema, grad, numel, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), ema_decay,
      grad_scale);
  return gg::launch_status("adam_ema");
}
