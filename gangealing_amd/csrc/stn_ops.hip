// a7-a10  the small STN-side operators that the reference spreads over ~40 tiny torch launches:
//   affine_grid (+bwd), RAFT convex flow upsampling + identity + affine composition (+bwd),
//   bilinear flow resize (+bwd), BilinearDownsample (+bwd), TV / identity flow losses (+bwd).
// All of them are latency bound at batch 16 (<= 10 MB of traffic), so each is one kernel per
// direction (two for the scatter-shaped backward passes), thread-per-output.  No floating-point atomics: grid-wide
// sums are finished in a fixed order by the last block (gg::ordered_grid_sum) and scatter-shaped gradients are
// written per source and then GATHERED per destination, so every result is bitwise reproducible.
#include "../../include/gangealing_hip.h"
#include "gg_common.h"

namespace {

// torch.linspace(-1, 1, steps)[i] * (steps-1)/steps  - base coordinate of F.affine_grid(align_corners=False)
__device__ __forceinline__ float base_coord(int i, int steps) {
  if (steps <= 1) return 0.f;
  const float step = 2.f / (float)(steps - 1);
  const float lin = (i < steps / 2) ? (-1.f + step * (float)i) : (1.f - step * (float)(steps - 1 - i));
  return lin * (float)(steps - 1) / (float)steps;
}

// ---------------------------------------------------------------- affine_grid

__global__ __launch_bounds__(256) void affine_grid_kernel(float* __restrict__ grid, const float* __restrict__ theta,
                                                          long long total, int ho, int wo) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const int j = (int)(o % wo);
    const long long q = o / wo;
    const int i = (int)(q % ho);
    const float* t = theta + (q / ho) * 6;
    const float x = base_coord(j, wo), y = base_coord(i, ho);
    grid[o * 2 + 0] = x * t[0] + y * t[1] + t[2];
    grid[o * 2 + 1] = x * t[3] + y * t[4] + t[5];
  }
}

// grid = (splits, n); each block reduces its share of one sample's pixels to 6 numbers.
__global__ __launch_bounds__(256) void affine_grid_bwd_kernel(float* __restrict__ gtheta,
                                                              const float* __restrict__ ggrid, int ho, int wo,
                                                              float* part, unsigned* ticket) {
  __shared__ float red[4];
  const int s = blockIdx.y;
  const long long pix = (long long)ho * wo;
  const float* g = ggrid + (size_t)s * pix * 2;
  float acc[6] = {0, 0, 0, 0, 0, 0};
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < pix; p += (long long)gridDim.x * 256) {
    const int j = (int)(p % wo), i = (int)(p / wo);
    const float x = base_coord(j, wo), y = base_coord(i, ho);
    const float gx = g[p * 2], gy = g[p * 2 + 1];
    acc[0] += gx * x; acc[1] += gx * y; acc[2] += gx;
    acc[3] += gy * x; acc[4] += gy * y; acc[5] += gy;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) acc[k] = gg::block_sum_256<float>(acc[k], red);
  if (gg::ordered_grid_sum<float, 6>(acc, part, ticket, s, blockIdx.x, gridDim.x, red)) {
#pragma unroll
    for (int k = 0; k < 6; ++k) gtheta[s * 6 + k] = acc[k];
  }
}

// ---------------------------------------------------------------- similarity matrix
// SimilarityHead.make_affine_matrix (warping_heads.py:36-56): the regressed (N, 4K) row [rot | log-scale | shift_x |
// shift_y] (K heads each) -> K matrices [[s cos r, -s sin r, tx], [s sin r, s cos r, ty]], r = pi tanh(rot), s = exp(.).
// The reference spends 11 element-wise launches on (N, K) tensors here and ~25 more in the backward; one thread per
// (sample, head) does either direction.  Same operations in the same order as the ATen kernels the reference runs
// (tanh, * float(pi), exp, cos, sin, products - nothing to contract).
constexpr float kPiF = 3.14159274101257324f;         // float(math.pi), the scalar ATen multiplies by

__global__ __launch_bounds__(256) void similarity_matrix_kernel(float* __restrict__ m, const float* __restrict__ p,
                                                                int n, int k) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * k) return;
  const int i = idx / k, h = idx - i * k;
  const float* row = p + (size_t)i * 4 * k;
  const float r = gg::mul_rn(tanhf(row[h]), kPiF);
  const float s = expf(row[k + h]);
  const float c = cosf(r), sn = sinf(r);
  float* o = m + (size_t)idx * 6;
  o[0] = gg::mul_rn(s, c);
  o[1] = gg::mul_rn(-s, sn);
  o[2] = row[2 * k + h];
  o[3] = gg::mul_rn(s, sn);
  o[4] = gg::mul_rn(s, c);
  o[5] = row[3 * k + h];
}

__global__ __launch_bounds__(256) void similarity_matrix_bwd_kernel(float* __restrict__ gp, const float* __restrict__ gm,
                                                                    const float* __restrict__ p, int n, int k) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * k) return;
  const int i = idx / k, h = idx - i * k;
  const float* row = p + (size_t)i * 4 * k;
  const float t = tanhf(row[h]);
  const float r = gg::mul_rn(t, kPiF);
  const float s = expf(row[k + h]);
  const float c = cosf(r), sn = sinf(r);
  const float* g = gm + (size_t)idx * 6;
  const float ga = g[0] + g[4];                 // d / d(s cos r)
  const float gb = g[3] - g[1];                 // d / d(s sin r)
  const float gs = ga * c + gb * sn;
  const float gr = s * (gb * c - ga * sn);
  float* o = gp + (size_t)i * 4 * k;
  o[h] = gr * kPiF * (1.f - t * t);
  o[k + h] = gs * s;
  o[2 * k + h] = g[2];
  o[3 * k + h] = g[5];
}

// ---------------------------------------------------------------- flow composition

struct Convex {
  float s[9];        // softmax weights
  float vx[9], vy[9];  // ds * neighbour flow
};

__device__ __forceinline__ Convex convex_load(const float* __restrict__ low, const float* __restrict__ mask, int n,
                                              int ij, int y, int x, int hl, int wl, int ds) {
  Convex c;
  const int dd = ds * ds;
  const size_t plane = (size_t)hl * wl;
  const float* m = mask + ((size_t)n * 9 * dd + ij) * plane + (size_t)y * wl + x;
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    c.s[k] = m[(size_t)k * dd * plane];
    mx = fmaxf(mx, c.s[k]);
  }
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    c.s[k] = expf(c.s[k] - mx);
    sum += c.s[k];
  }
  const float inv = 1.f / sum;
  const float* lx = low + (size_t)n * 2 * plane;
  const float* ly = lx + plane;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    c.s[k] *= inv;
    const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
    const bool in = yy >= 0 && yy < hl && xx >= 0 && xx < wl;
    c.vx[k] = in ? (float)ds * lx[(size_t)yy * wl + xx] : 0.f;
    c.vy[k] = in ? (float)ds * ly[(size_t)yy * wl + xx] : 0.f;
  }
  return c;
}

// thread -> (ij, y, x) with x fastest so that mask reads are coalesced along x; blockIdx.y = sample
__global__ __launch_bounds__(256) void flow_compose_fwd_kernel(float* __restrict__ delta, float* __restrict__ flow,
                                                               const float* __restrict__ low,
                                                               const float* __restrict__ mask,
                                                               const float* __restrict__ base, int hl, int wl,
                                                               int ds) {
  const int n = blockIdx.y;
  const int per = ds * ds * hl * wl;
  const int H = ds * hl, W = ds * wl;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < per; t += gridDim.x * 256) {
    const int x = t % wl;
    const int y = (t / wl) % hl;
    const int ij = t / (wl * hl);
    const Convex c = convex_load(low, mask, n, ij, y, x, hl, wl, ds);
    float ux = 0.f, uy = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { ux += c.s[k] * c.vx[k]; uy += c.s[k] * c.vy[k]; }
    const int Y = ds * y + ij / ds, X = ds * x + ij % ds;
    const size_t o = (((size_t)n * H + Y) * W + X) * 2;
    delta[o] = ux;
    delta[o + 1] = uy;
    float fx = base_coord(X, W) + ux, fy = base_coord(Y, H) + uy;
    if (base) {
      const float* b = base + n * 6;
      const float ax = fx * b[0] + fy * b[1] + b[2];
      const float ay = fx * b[3] + fy * b[4] + b[5];
      fx = ax; fy = ay;
    }
    flow[o] = fx;
    flow[o + 1] = fy;
  }
}

// contrib: per (n, k, ij, y, x) the share ds * s[k] * (gux, guy) that hi-res pixel (ij, y, x) sends to its low-res
// neighbour k (same index layout as gmask); flow_compose_gather_kernel sums them per low-res pixel in a fixed order
__global__ __launch_bounds__(256) void flow_compose_bwd_kernel(
    float2* __restrict__ contrib, float* __restrict__ gmask, float* __restrict__ gbase, const float* __restrict__ g_flow,
    const float* __restrict__ g_delta, const float* __restrict__ low, const float* __restrict__ mask,
    const float* __restrict__ base, int hl, int wl, int ds, float* part, unsigned* ticket) {
  __shared__ float red[4];
  const int n = blockIdx.y;
  const int per = ds * ds * hl * wl;
  const int H = ds * hl, W = ds * wl;
  const int dd = ds * ds;
  const size_t plane = (size_t)hl * wl;
  float accb[6] = {0, 0, 0, 0, 0, 0};
  const int iters = (per + gridDim.x * 256 - 1) / (gridDim.x * 256);
  for (int it = 0; it < iters; ++it) {
    const int t = (it * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
    if (t >= per) continue;
    const int x = t % wl;
    const int y = (t / wl) % hl;
    const int ij = t / (wl * hl);
    const Convex c = convex_load(low, mask, n, ij, y, x, hl, wl, ds);
    const int Y = ds * y + ij / ds, X = ds * x + ij % ds;
    const size_t o = (((size_t)n * H + Y) * W + X) * 2;
    float gfx = 0.f, gfy = 0.f, gux = 0.f, guy = 0.f;
    if (g_flow) { gfx = g_flow[o]; gfy = g_flow[o + 1]; }
    if (g_delta) { gux = g_delta[o]; guy = g_delta[o + 1]; }
    if (base) {
      const float* b = base + n * 6;
      float ux = 0.f, uy = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) { ux += c.s[k] * c.vx[k]; uy += c.s[k] * c.vy[k]; }
      const float fx = base_coord(X, W) + ux, fy = base_coord(Y, H) + uy;
      accb[0] += gfx * fx; accb[1] += gfx * fy; accb[2] += gfx;
      accb[3] += gfy * fx; accb[4] += gfy * fy; accb[5] += gfy;
      gux += b[0] * gfx + b[3] * gfy;
      guy += b[1] * gfx + b[4] * gfy;
    } else {
      gux += gfx;
      guy += gfy;
    }
    float gs[9], dot = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      gs[k] = gux * c.vx[k] + guy * c.vy[k];
      dot += c.s[k] * gs[k];
    }
    const size_t e0 = ((size_t)n * 9 * dd + ij) * plane + (size_t)y * wl + x;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      gmask[e0 + (size_t)k * dd * plane] = c.s[k] * (gs[k] - dot);
      contrib[e0 + (size_t)k * dd * plane] = make_float2((float)ds * c.s[k] * gux, (float)ds * c.s[k] * guy);
    }
  }
  if (base && gbase) {
#pragma unroll
    for (int k = 0; k < 6; ++k) accb[k] = gg::block_sum_256<float>(accb[k], red);
    if (gg::ordered_grid_sum<float, 6>(accb, part, ticket, n, blockIdx.x, gridDim.x, red)) {
#pragma unroll
      for (int k = 0; k < 6; ++k) gbase[n * 6 + k] = accb[k];
    }
  }
}

// glow[n][c][yy][xx] = sum over the 9 neighbour slots k and the ds*ds sub-positions ij of the shares aimed at
// (yy, xx).  One wave per low-res pixel: lane = ij (strided when ds*ds > 64), k ascending, then the wave's fixed
// shuffle tree - a fixed order.  grid = (ceil(hl*wl / 4), n), 4 waves per block.
__global__ __launch_bounds__(256) void flow_compose_gather_kernel(float* __restrict__ glow,
                                                                  const float2* __restrict__ contrib, int hl, int wl,
                                                                  int dd) {
  const int n = blockIdx.y, lane = threadIdx.x & 63;
  const int pix = blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t plane = (size_t)hl * wl;
  if (pix >= hl * wl) return;
  const int yy = pix / wl, xx = pix - yy * wl;
  float sx = 0.f, sy = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int y = yy - (k / 3 - 1), x = xx - (k % 3 - 1);          // the hi-res cell whose slot k is (yy, xx)
    if (y < 0 || y >= hl || x < 0 || x >= wl) continue;
    const float2* src = contrib + ((size_t)n * 9 + k) * dd * plane + (size_t)y * wl + x;
    for (int ij = lane; ij < dd; ij += 64) {
      const float2 v = src[(size_t)ij * plane];
      sx += v.x;
      sy += v.y;
    }
  }
  sx = gg::wave_sum(sx);
  sy = gg::wave_sum(sy);
  if (lane == 0) {
    glow[(size_t)n * 2 * plane + pix] = sx;
    glow[(size_t)n * 2 * plane + plane + pix] = sy;
  }
}

// ---------------------------------------------------------------- bilinear flow resize

struct Lin { int i0, i1; float l0, l1; };

__device__ __forceinline__ Lin lin_src(int dst, float rscale, int size) {   // ATen UpSample.h, align_corners=False
  float src = rscale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  Lin l;
  l.i0 = min((int)src, size - 1);
  l.i1 = l.i0 + (l.i0 < size - 1 ? 1 : 0);
  l.l1 = src - (float)l.i0;
  l.l0 = 1.f - l.l1;
  return l;
}

__global__ __launch_bounds__(256) void flow_resize_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                          long long total, int hi, int wi, int ho, int wo,
                                                          float rscale) {
  // dst = out (N,ho,wo,2), src = in (N,hi,wi,2)
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const int x = (int)(o % wo);
    const long long q = o / wo;
    const int y = (int)(q % ho);
    const size_t n = (size_t)(q / ho);
    const Lin ly = lin_src(y, rscale, hi), lx = lin_src(x, rscale, wi);
    const size_t b = n * hi * wi;
    const size_t p00 = (b + (size_t)ly.i0 * wi + lx.i0) * 2, p01 = (b + (size_t)ly.i0 * wi + lx.i1) * 2;
    const size_t p10 = (b + (size_t)ly.i1 * wi + lx.i0) * 2, p11 = (b + (size_t)ly.i1 * wi + lx.i1) * 2;
#pragma unroll
    for (int c = 0; c < 2; ++c)
      dst[o * 2 + c] = ly.l0 * (lx.l0 * src[p00 + c] + lx.l1 * src[p01 + c]) +
                       ly.l1 * (lx.l0 * src[p10 + c] + lx.l1 * src[p11 + c]);
  }
}

// Outputs whose interpolation footprint along one axis contains input index i: a conservative index range (the
// footprint test itself is repeated with lin_src, so the weights are exactly the forward's).  src = rscale * (o + .5)
// - .5 lies in (i - 1, i + 1) for interior outputs; outputs clamped to src = 0 belong to i = 0.
__device__ __forceinline__ void resize_sources(int i, float rscale, int osize, int& lo, int& hi) {
  const float inv = 1.f / rscale;
  lo = (int)floorf(((float)i - 1.f + 0.5f) * inv - 0.5f) - 1;
  hi = (int)ceilf(((float)i + 1.f + 0.5f) * inv - 0.5f) + 1;
  if (i == 0) lo = 0;
  if (lo < 0) lo = 0;
  if (hi > osize - 1) hi = osize - 1;
}

// backward as a GATHER: thread per input pixel, visits the candidate outputs in index order (fixed summation order,
// no atomics); dst = grad_in (N,hi,wi,2), src = grad_out (N,ho,wo,2)
__global__ __launch_bounds__(256) void flow_resize_bwd_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                              long long total_in, int hi, int wi, int ho, int wo,
                                                              float rscale) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < total_in; p += stride) {
    const int xi = (int)(p % wi);
    const long long q = p / wi;
    const int yi = (int)(q % hi);
    const size_t n = (size_t)(q / hi);
    int ylo, yhi, xlo, xhi;
    resize_sources(yi, rscale, ho, ylo, yhi);
    resize_sources(xi, rscale, wo, xlo, xhi);
    float gx = 0.f, gy = 0.f;
    for (int y = ylo; y <= yhi; ++y) {
      const Lin ly = lin_src(y, rscale, hi);
      if (ly.i0 != yi && ly.i1 != yi) continue;
      const float* row = src + ((n * ho + y) * (size_t)wo) * 2;
      for (int x = xlo; x <= xhi; ++x) {
        const Lin lx = lin_src(x, rscale, wi);
        if (lx.i0 != xi && lx.i1 != xi) continue;
        // the forward's four products, kept separate so that every weight is formed exactly as there
        float ax = 0.f, ay = 0.f;
        if (ly.i0 == yi && lx.i0 == xi) { ax += row[x * 2] * ly.l0 * lx.l0; ay += row[x * 2 + 1] * ly.l0 * lx.l0; }
        if (ly.i0 == yi && lx.i1 == xi) { ax += row[x * 2] * ly.l0 * lx.l1; ay += row[x * 2 + 1] * ly.l0 * lx.l1; }
        if (ly.i1 == yi && lx.i0 == xi) { ax += row[x * 2] * ly.l1 * lx.l0; ay += row[x * 2 + 1] * ly.l1 * lx.l0; }
        if (ly.i1 == yi && lx.i1 == xi) { ax += row[x * 2] * ly.l1 * lx.l1; ay += row[x * 2 + 1] * ly.l1 * lx.l1; }
        gx += ax;
        gy += ay;
      }
    }
    dst[p * 2] = gx;
    dst[p * 2 + 1] = gy;
  }
}

// ---------------------------------------------------------------- BilinearDownsample

constexpr int MAX_STRIDE = 8;

__device__ __forceinline__ int reflect_idx(int i, int n) {
  if (i < 0) return -i;
  if (i >= n) return 2 * n - 2 - i;
  return i;
}

__device__ __forceinline__ float tent(int t, int stride) {          // kernel[t], t in [0, 2*stride)
  const int v = (t < stride) ? 2 * t + 1 : 2 * (2 * stride - 1 - t) + 1;
  return (float)((double)v / (double)(2 * stride * stride));
}

constexpr int TENT_LDS = 64;
__global__ __launch_bounds__(256) void bilinear_down_kernel(float* __restrict__ out, const float* __restrict__ in,
                                                            long long total, int h, int w, int stride) {
  const int r = stride / 2;
  const int oh = (h + 2 * r - 2 * stride) / stride + 1, ow = (w + 2 * r - 2 * stride) / stride + 1;
  // the 2 * stride taps once per block (tent() divides in double: exact, and ~20 of them per output in the loops below)
  __shared__ float tw[TENT_LDS];
  const bool tabled = 2 * stride <= TENT_LDS;
  if (tabled && (int)threadIdx.x < 2 * stride) tw[threadIdx.x] = tent((int)threadIdx.x, stride);
  __syncthreads();
  const long long gstride = (long long)gridDim.x * blockDim.x;
  const bool small = total < (1LL << 31);
  for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += gstride) {
    int ox, oy;
    long long plane;
    if (small) {
      const unsigned q = (unsigned)o / (unsigned)ow, pl = q / (unsigned)oh;
      ox = (int)((unsigned)o - q * (unsigned)ow); oy = (int)(q - pl * (unsigned)oh); plane = pl;
    } else {
      ox = (int)(o % ow);
      const long long q = o / ow;
      oy = (int)(q % oh); plane = q / oh;
    }
    const float* src = in + (size_t)plane * h * w;
    float acc = 0.f;
    for (int ty = 0; ty < 2 * stride; ++ty) {
      const int iy = reflect_idx(stride * oy + ty - r, h);
      float row = 0.f;
      for (int tx = 0; tx < 2 * stride; ++tx) {
        const int ix = reflect_idx(stride * ox + tx - r, w);
        row += src[(size_t)iy * w + ix] * (tabled ? tw[tx] : tent(tx, stride));
      }
      acc += row * (tabled ? tw[ty] : tent(ty, stride));
    }
    out[o] = acc;
  }
}

// padded positions q whose source is input index i (ReflectionPad(r)): q = i + r, plus mirror images
__device__ __forceinline__ int pad_sources(int i, int n, int r, int q[3]) {
  int cnt = 0;
  q[cnt++] = i + r;
  if (i >= 1 && i <= r) q[cnt++] = r - i;
  if (i >= n - 1 - r && i <= n - 2) q[cnt++] = 2 * n - 2 - i + r;
  return cnt;
}

__global__ __launch_bounds__(256) void bilinear_down_bwd_kernel(float* __restrict__ gin,
                                                                const float* __restrict__ gout, long long total,
                                                                int h, int w, int stride) {
  const int r = stride / 2;
  const int oh = (h + 2 * r - 2 * stride) / stride + 1, ow = (w + 2 * r - 2 * stride) / stride + 1;
  __shared__ float tw[TENT_LDS];                            // (as in the forward kernel)
  const bool tabled = 2 * stride <= TENT_LDS;
  if (tabled && (int)threadIdx.x < 2 * stride) tw[threadIdx.x] = tent((int)threadIdx.x, stride);
  __syncthreads();
  const long long gstride = (long long)gridDim.x * blockDim.x;
  const bool small = total < (1LL << 31);                  // 32-bit index math (a 64-bit division is ~100 VALU instructions)
  for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += gstride) {
    int ix, iy;
    long long plane;
    if (small) {
      const unsigned q = (unsigned)o / (unsigned)w, pl = q / (unsigned)h;
      ix = (int)((unsigned)o - q * (unsigned)w); iy = (int)(q - pl * (unsigned)h); plane = pl;
    } else {
      ix = (int)(o % w);
      const long long q = o / w;
      iy = (int)(q % h); plane = q / h;
    }
    const float* g = gout + (size_t)plane * oh * ow;
    int qy[3], qx[3];
    const int ny = pad_sources(iy, h, r, qy), nx = pad_sources(ix, w, r, qx);
    float acc = 0.f;
    for (int a = 0; a < ny; ++a) {
      const int y_hi = min(qy[a] / stride, oh - 1);
      const int y_lo = max((qy[a] - 2 * stride + 1 + stride - 1) / stride, 0);     // ceil((q-2s+1)/s), q-2s+1 may be < 0
      for (int oy = max(y_lo, 0); oy <= y_hi; ++oy) {
        const int ty = qy[a] - stride * oy;
        if (ty < 0 || ty >= 2 * stride) continue;
        const float ky = tabled ? tw[ty] : tent(ty, stride);
        for (int b = 0; b < nx; ++b) {
          const int x_hi = min(qx[b] / stride, ow - 1);
          for (int ox = max(x_hi - 2, 0); ox <= x_hi; ++ox) {
            const int tx = qx[b] - stride * ox;
            if (tx < 0 || tx >= 2 * stride) continue;
            acc += g[(size_t)oy * ow + ox] * ky * (tabled ? tw[tx] : tent(tx, stride));
          }
        }
      }
    }
    gin[o] = acc;
  }
}

// ---------------------------------------------------------------- flow losses

__device__ __forceinline__ float huber(float u) {
  const float a = fabsf(u);
  return a <= 1.f ? 0.5f * a * a : a - 0.5f;
}
__device__ __forceinline__ float huber_grad(float u) {
  return fabsf(u) <= 1.f ? u : (u > 0.f ? 1.f : -1.f);
}

__global__ __launch_bounds__(256) void flow_losses_kernel(float* __restrict__ losses, const float* __restrict__ d,
                                                          long long total, int hf, int wf, float inv_y, float inv_x,
                                                          float inv_all, float* part, unsigned* ticket) {
  __shared__ float red[4];
  float sy = 0.f, sx = 0.f, sq = 0.f;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const long long pix = o >> 1;
    const int x = (int)(pix % wf);
    const int y = (int)((pix / wf) % hf);
    const float v = d[o];
    sq += v * v;
    if (y + 1 < hf) sy += huber(v - d[o + (size_t)wf * 2]);
    if (x + 1 < wf) sx += huber(v - d[o + 2]);
  }
  const float ty = gg::block_sum_256<float>(sy, red);
  const float tx = gg::block_sum_256<float>(sx, red);
  const float tq = gg::block_sum_256<float>(sq, red);
  float v[2] = {ty * inv_y + tx * inv_x, tq * inv_all};
  if (gg::ordered_grid_sum<float, 2>(v, part, ticket, 0, blockIdx.x, gridDim.x, red)) {
    losses[0] = v[0];
    losses[1] = v[1];
  }
}

__global__ __launch_bounds__(256) void flow_losses_bwd_kernel(float* __restrict__ gd, const float* __restrict__ d,
                                                              const float* __restrict__ g_losses, long long total,
                                                              int hf, int wf, float inv_y, float inv_x,
                                                              float inv_all) {
  const float g_tv = g_losses[0], g_id = g_losses[1];
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const long long pix = o >> 1;
    const int x = (int)(pix % wf);
    const int y = (int)((pix / wf) % hf);
    const float v = d[o];
    float g = 0.f;
    if (y + 1 < hf) g += huber_grad(v - d[o + (size_t)wf * 2]) * inv_y;
    if (y > 0) g -= huber_grad(d[o - (size_t)wf * 2] - v) * inv_y;
    if (x + 1 < wf) g += huber_grad(v - d[o + 2]) * inv_x;
    if (x > 0) g -= huber_grad(d[o - 2] - v) * inv_x;
    gd[o] = g_tv * g + g_id * 2.f * v * inv_all;
  }
}

}  // namespace

extern "C" int gg_similarity_matrix_f32(float* matrix, const float* params, int n, int heads, void* stream) {
  if (n <= 0 || heads <= 0) return 0;
  if (!matrix || !params) return gg::fail(-2, "similarity_matrix: null pointer");
  if ((long long)n * heads > (1LL << 30)) return gg::fail(-2, "similarity_matrix: too many matrices");
  similarity_matrix_kernel<<<(unsigned)((n * heads + 255) / 256), 256, 0, gg::as_stream(stream)>>>(matrix, params, n, heads);
  return gg::launch_status("similarity_matrix");
}

extern "C" int gg_similarity_matrix_bwd_f32(float* grad_params, const float* grad_matrix, const float* params, int n,
                                            int heads, void* stream) {
  if (n <= 0 || heads <= 0) return 0;
  if (!grad_params || !grad_matrix || !params) return gg::fail(-2, "similarity_matrix_bwd: null pointer");
  if ((long long)n * heads > (1LL << 30)) return gg::fail(-2, "similarity_matrix_bwd: too many matrices");
  similarity_matrix_bwd_kernel<<<(unsigned)((n * heads + 255) / 256), 256, 0, gg::as_stream(stream)>>>(
      grad_params, grad_matrix, params, n, heads);
  return gg::launch_status("similarity_matrix_bwd");
}

extern "C" int gg_affine_grid_f32(float* grid, const float* theta, int n, int ho, int wo, void* stream) {
  const long long total = (long long)n * ho * wo;
  if (total <= 0) return 0;
  if (!grid || !theta) return gg::fail(-2, "affine_grid: null pointer");
  affine_grid_kernel<<<gg::stream_grid(total, 256), 256, 0, gg::as_stream(stream)>>>(grid, theta, total, ho, wo);
  return gg::launch_status("affine_grid");
}

extern "C" int gg_affine_grid_bwd_f32(float* grad_theta, const float* grad_grid, int n, int ho, int wo,
                                      void* stream) {
  if (n <= 0) return 0;
  if (!grad_theta || !grad_grid || n > 65535) return gg::fail(-2, "affine_grid_bwd: bad arguments");
  hipStream_t st = gg::as_stream(stream);
  const long long pix = (long long)ho * wo;
  if (pix <= 0) {
    hipError_t e = hipMemsetAsync(grad_theta, 0, sizeof(float) * 6 * (size_t)n, st);
    return e == hipSuccess ? 0 : gg::fail((int)e, "affine_grid_bwd: memset failed");
  }
  unsigned splits = (unsigned)((pix + 4095) / 4096);
  if (splits > 64) splits = 64;
  if (n > gg::kTickets) splits = 1;
  float* part = nullptr;
  unsigned* ticket = nullptr;
  if (splits > 1) {
    part = reinterpret_cast<float*>(gg::scratch(st, sizeof(float) * 6 * (size_t)n * splits));
    ticket = gg::tickets(st);
    if (!part || !ticket) return -3;
  }
  affine_grid_bwd_kernel<<<dim3(splits, n), 256, 0, st>>>(grad_theta, grad_grid, ho, wo, part, ticket);
  return gg::launch_status("affine_grid_bwd");
}

extern "C" int gg_flow_compose_fwd_f32(float* delta, float* flow, const float* low_flow, const float* mask,
                                       const float* base, int n, int hl, int wl, int ds, void* stream) {
  if (n <= 0 || hl <= 0 || wl <= 0) return 0;
  if (!delta || !flow || !low_flow || !mask || ds < 1 || n > 65535) return gg::fail(-2, "flow_compose_fwd: bad arguments");
  const long long per = (long long)ds * ds * hl * wl;
  unsigned bx = (unsigned)((per + 255) / 256);
  if (bx > 1024) bx = 1024;
  flow_compose_fwd_kernel<<<dim3(bx, n), 256, 0, gg::as_stream(stream)>>>(delta, flow, low_flow, mask, base, hl, wl,
                                                                         ds);
  return gg::launch_status("flow_compose_fwd");
}

extern "C" int gg_flow_compose_bwd_f32(float* grad_low, float* grad_mask, float* grad_base, const float* g_flow,
                                       const float* g_delta, const float* low_flow, const float* mask,
                                       const float* base, int n, int hl, int wl, int ds, void* stream) {
  if (n <= 0 || hl <= 0 || wl <= 0) return 0;
  if (!grad_low || !grad_mask || !low_flow || !mask || ds < 1 || n > 65535)
    return gg::fail(-2, "flow_compose_bwd: bad arguments");
  hipStream_t st = gg::as_stream(stream);
  const long long per = (long long)ds * ds * hl * wl;
  unsigned bx = (unsigned)((per + 255) / 256);
  if (bx > 1024) bx = 1024;
  if (n > gg::kTickets) return gg::fail(-2, "flow_compose_bwd: batch too large");
  // scratch: [contributions: n * 9 * ds^2 * hl * wl float2][partials of the base-matrix gradient: n * bx * 6]
  const size_t contrib_bytes = sizeof(float2) * (size_t)n * 9 * per;
  char* sc = reinterpret_cast<char*>(gg::scratch(st, contrib_bytes + sizeof(float) * 6 * (size_t)n * bx));
  unsigned* ticket = gg::tickets(st);
  if (!sc || !ticket) return -3;
  float2* contrib = reinterpret_cast<float2*>(sc);
  float* part = reinterpret_cast<float*>(sc + contrib_bytes);
  flow_compose_bwd_kernel<<<dim3(bx, n), 256, 0, st>>>(contrib, grad_mask, grad_base, g_flow, g_delta, low_flow, mask,
                                                       base, hl, wl, ds, part, ticket);
  int rc = gg::launch_status("flow_compose_bwd");
  if (rc) return rc;
  flow_compose_gather_kernel<<<dim3((unsigned)((hl * wl + 3) / 4), n), 256, 0, st>>>(grad_low, contrib, hl, wl,
                                                                                    ds * ds);
  return gg::launch_status("flow_compose_gather");
}

extern "C" int gg_flow_resize_f32(float* out, const float* in, int n, int hi, int wi, int ho, int wo, float scale,
                                  void* stream) {
  const long long total = (long long)n * ho * wo;
  if (total <= 0) return 0;
  if (!out || !in || hi <= 0 || wi <= 0 || !(scale > 0.f)) return gg::fail(-2, "flow_resize: bad arguments");
  flow_resize_kernel<<<gg::stream_grid(total, 256), 256, 0, gg::as_stream(stream)>>>(out, in, total, hi, wi, ho, wo,
                                                                                    1.f / scale);
  return gg::launch_status("flow_resize");
}

extern "C" int gg_flow_resize_bwd_f32(float* grad_in, const float* grad_out, int n, int hi, int wi, int ho, int wo,
                                      float scale, void* stream) {
  const long long total = (long long)n * ho * wo;
  if (n <= 0 || hi <= 0 || wi <= 0) return 0;
  if (!grad_in || !grad_out || !(scale > 0.f)) return gg::fail(-2, "flow_resize_bwd: bad arguments");
  hipStream_t st = gg::as_stream(stream);
  if (total <= 0) {
    hipError_t e = hipMemsetAsync(grad_in, 0, sizeof(float) * 2 * (size_t)n * hi * wi, st);
    return e == hipSuccess ? 0 : gg::fail((int)e, "flow_resize_bwd: memset failed");
  }
  const long long total_in = (long long)n * hi * wi;
  flow_resize_bwd_kernel<<<gg::stream_grid(total_in, 256), 256, 0, st>>>(grad_in, grad_out, total_in, hi, wi, ho, wo,
                                                                         1.f / scale);
  return gg::launch_status("flow_resize_bwd");
}

extern "C" int gg_bilinear_downsample_f32(float* out, const float* in, int planes, int h, int w, int stride,
                                          void* stream) {
  if (planes <= 0) return 0;
  if (!out || !in || stride < 1 || stride > MAX_STRIDE || h <= stride / 2 || w <= stride / 2)
    return gg::fail(-2, "bilinear_downsample: bad arguments");
  const int r = stride / 2;
  const int oh = (h + 2 * r - 2 * stride) / stride + 1, ow = (w + 2 * r - 2 * stride) / stride + 1;
  const long long total = (long long)planes * oh * ow;
  if (total <= 0) return 0;
  bilinear_down_kernel<<<gg::stream_grid(total, 256), 256, 0, gg::as_stream(stream)>>>(out, in, total, h, w, stride);
  return gg::launch_status("bilinear_downsample");
}

extern "C" int gg_bilinear_downsample_bwd_f32(float* grad_in, const float* grad_out, int planes, int h, int w,
                                              int stride, void* stream) {
  if (planes <= 0) return 0;
  if (!grad_in || !grad_out || stride < 1 || stride > MAX_STRIDE || h <= stride / 2 || w <= stride / 2)
    return gg::fail(-2, "bilinear_downsample_bwd: bad arguments");
  const long long total = (long long)planes * h * w;
  bilinear_down_bwd_kernel<<<gg::stream_grid(total, 256), 256, 0, gg::as_stream(stream)>>>(grad_in, grad_out, total,
                                                                                          h, w, stride);
  return gg::launch_status("bilinear_downsample_bwd");
}

extern "C" int gg_flow_losses_f32(float* losses, const float* delta, int n, int hf, int wf, void* stream) {
  if (!losses) return gg::fail(-2, "flow_losses: null pointer");
  hipStream_t st = gg::as_stream(stream);
  const long long total = (long long)n * hf * wf * 2;
  if (total <= 0) {
    hipError_t e = hipMemsetAsync(losses, 0, sizeof(float) * 2, st);
    return e == hipSuccess ? 0 : gg::fail((int)e, "flow_losses: memset failed");
  }
  if (!delta) return gg::fail(-2, "flow_losses: null pointer");
  const float inv_y = hf > 1 ? 1.f / (float)((long long)n * (hf - 1) * wf * 2) : 0.f;
  const float inv_x = wf > 1 ? 1.f / (float)((long long)n * hf * (wf - 1) * 2) : 0.f;
  unsigned blocks = gg::stream_grid(total, 256);
  if (blocks > 256) blocks = 256;
  float* part = nullptr;
  unsigned* ticket = nullptr;
  if (blocks > 1) {
    part = reinterpret_cast<float*>(gg::scratch(st, sizeof(float) * 2 * blocks));
    ticket = gg::tickets(st);
    if (!part || !ticket) return -3;
  }
  flow_losses_kernel<<<blocks, 256, 0, st>>>(losses, delta, total, hf, wf, inv_y, inv_x, 1.f / (float)total, part,
                                             ticket);
  return gg::launch_status("flow_losses");
}

extern "C" int gg_flow_losses_bwd_f32(float* grad_delta, const float* delta, const float* g_losses, int n, int hf,
                                      int wf, void* stream) {
  const long long total = (long long)n * hf * wf * 2;
  if (total <= 0) return 0;
  if (!grad_delta || !delta || !g_losses) return gg::fail(-2, "flow_losses_bwd: null pointer");
  const float inv_y = hf > 1 ? 1.f / (float)((long long)n * (hf - 1) * wf * 2) : 0.f;
  const float inv_x = wf > 1 ? 1.f / (float)((long long)n * hf * (wf - 1) * 2) : 0.f;
  flow_losses_bwd_kernel<<<gg::stream_grid(total, 256), 256, 0, gg::as_stream(stream)>>>(
      grad_delta, delta, g_losses, total, hf, wf, inv_y, inv_x, 1.f / (float)total);
  return gg::launch_status("flow_losses_bwd");
}
