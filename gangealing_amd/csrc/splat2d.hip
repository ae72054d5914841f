// a11  splat2d: Gaussian soft-scatter of (x, y, value) points (forward only, as in the reference).
//
// Semantics follow utils/splat2d_cuda/src/splat_gpu_impl.cu:41-96 and splat_gpu.c:12-42:
//   points outside [0,W) x [0,H) are dropped; the footprint is the +-2*sigma box clipped to the
//   image; weight = exp(-((px-x)^2 + (py-y)^2) / (2 sigma^2)); alpha and alpha*value are
//   accumulated per pixel; the host glue then divides by (alpha [clamped >= 1] + 1e-8).
//
// MI355X mapping (round 4): off the atomic unit.  The reference - and rounds 1-3 here - issue one floating-point
// atomic per (point, pixel, channel): 1e6 points at sigma 1.3 are 2e8 read-modify-writes that the L2 atomic units
// serialise (measured: 1.18 GB of memory-side write traffic for 32-58 MB of result, 14-80 GB/s of useful bytes).
// Now the scatter is turned into a gather:
//   1. bin     every point goes into the list of each 32 x 32-pixel tile its footprint box overlaps (at most 2 x 2
//              tiles for boxes up to 33 pixels, i.e. sigma <= 8): an integer counting pass, a scan, a fill pass;
//   2. sort    each tile's list by point index (bitonic, in LDS up to 8192 entries): the fill pass's integer atomics
//              hand out slots in arrival order, the sort makes the summation order the point order - results are
//              bitwise reproducible, which the atomic formulation (and the reference) are not;
//   3. gather  one block per tile: thread = 4 consecutive pixels of one row, accumulators in registers; the tile's
//              points are staged through LDS 256 at a time and every wave skips the points whose rows it does not
//              own (scalar branch).  Each pixel is written ONCE, with the normalisation of splat_gpu.c:33-40 fused
//              into that write (gg_splat2d_f32).
// Boxes larger than 33 pixels (sigma > 8: a point would enter up to (box / 32 + 1)^2 lists) take the round-3 path -
// 16 lanes per point, hardware float atomics - before the gather, which adds onto what they left.
#include "../../include/gangealing_hip.h"
#include "gg_common.h"

namespace {

constexpr int TS = 32;                 // tile edge in pixels
constexpr int kMaxBinBox = TS + 1;     // largest box edge that still overlaps at most two tiles per axis
constexpr int kSortLds = 8192;         // list entries sorted in LDS

struct Footprint {
  int t, b, l, r;
  float xc, yc, normalizer;
};

// the reference's box: splat_gpu_impl.cu:74-81
__device__ __forceinline__ bool footprint_xy(float xc, float yc, float stdev, int height, int width, Footprint& f) {
  f.xc = xc;
  f.yc = yc;
  if (!(f.xc >= 0.f && f.xc < (float)width && f.yc >= 0.f && f.yc < (float)height)) return false;
  const float length = 2.f * stdev;
  f.normalizer = -(1.f / (2.f * stdev * stdev));
  f.t = (int)fmaxf(0.f, floorf(f.yc - length));
  f.b = (int)fminf((float)(height - 1), ceilf(f.yc + length));
  f.l = (int)fmaxf(0.f, floorf(f.xc - length));
  f.r = (int)fminf((float)(width - 1), ceilf(f.xc + length));
  return f.r >= f.l && f.b >= f.t;
}
__device__ __forceinline__ bool footprint(const float* __restrict__ coords, const float* __restrict__ sigma,
                                          long long index, int num_points, int height, int width, Footprint& f) {
  return footprint_xy(coords[2 * (size_t)index], coords[2 * (size_t)index + 1], sigma[(int)(index / num_points)], height,
                      width, f);
}
__device__ __forceinline__ bool binned(const Footprint& f) {
  return f.r - f.l + 1 <= kMaxBinBox && f.b - f.t + 1 <= kMaxBinBox;
}

// ---- large boxes: 16 lanes per point, float atomics (the round-3 kernel, restricted to what the bins do not take) ----
__global__ __launch_bounds__(256) void splat_large_kernel(
    const float* __restrict__ coords, const float* __restrict__ values, const float* __restrict__ sigma,
    float* __restrict__ alpha_splats, float* __restrict__ output, int num_points, int channels, int height,
    int width, int top_count) {
  constexpr int G = 16;                                   // lanes per point
  const int sub = threadIdx.x & (G - 1);
  const long long group = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const long long ngroups = ((long long)gridDim.x * blockDim.x) / G;
  const size_t hw = (size_t)height * width;
  for (long long index = group; index < top_count; index += ngroups) {
    Footprint f;
    if (!footprint(coords, sigma, index, num_points, height, width, f) || binned(f)) continue;
    const int n = (int)(index / num_points);
    const int bw = f.r - f.l + 1, bh = f.b - f.t + 1;
    const int area = bw * bh;
    const float* val = values + (size_t)index * channels;
    float* a_img = alpha_splats + (size_t)n * hw;
    float* o_img = output + (size_t)n * channels * hw;
    for (int p = sub; p < area; p += G) {
      const int dy = p / bw, dx = p - dy * bw;
      const int lh = f.t + dy, lw = f.l + dx;
      const float fx = (float)lw - f.xc, fy = (float)lh - f.yc;
      const float alpha = expf(f.normalizer * (fx * fx + fy * fy));
      const size_t pix = (size_t)lh * width + lw;
      unsafeAtomicAdd(a_img + pix, alpha);
      for (int c = 0; c < channels; ++c) unsafeAtomicAdd(o_img + (size_t)c * hw + pix, alpha * val[c]);
    }
  }
}

// ---- binning: count / fill ----------------------------------------------------------------------------------------
// Integer atomics only (the COUNTS are order independent; the fill pass's slots are sorted afterwards).  One global atomic
// per (point, tile) was the first version - and the bottleneck: 1e6 points onto the 256 tile counters of a 512^2 image are
// ~4000 atomics per ADDRESS, which the L2 serialises (2.6 ms; the whole atomic-scatter kernel of round 3 took 0.7).  So a
// block aggregates kPtsPerBlock points of one image in an LDS histogram over the image's tiles and touches each global
// counter once.  FILL: the block reserves its range of every list with that one atomic and hands out the slots inside
// the range from LDS.  Images with more than kMaxLdsTiles tiles (beyond 2048^2) use direct atomics - there the
// contention per address is low anyway.
constexpr int kPtsPerBlock = 4096;
constexpr int kMaxLdsTiles = 4096;

template <bool FILL>
__global__ __launch_bounds__(256) void splat_bin_kernel(const float* __restrict__ coords, const float* __restrict__ sigma,
                                                        int* __restrict__ counts, const int* __restrict__ offsets,
                                                        int* __restrict__ lists, int num_points, int height, int width,
                                                        int top_count, int tiles_x, int tiles_y) {
  __shared__ int s_hist[kMaxLdsTiles];
  __shared__ int s_base[FILL ? kMaxLdsTiles : 1];
  const int n = blockIdx.y;
  const int tiles = tiles_x * tiles_y;
  const bool lds = tiles <= kMaxLdsTiles;
  const int p0 = blockIdx.x * kPtsPerBlock;
  int p1 = p0 + kPtsPerBlock;
  if (p1 > num_points) p1 = num_points;
  if ((long long)n * num_points + p1 > top_count) p1 = (int)(top_count - (long long)n * num_points);      // ragged last image
  int* gcount = counts + (size_t)n * tiles;
  if (lds) {
    for (int t = threadIdx.x; t < tiles; t += 256) s_hist[t] = 0;
    __syncthreads();
  }
  // pass 1: histogram of this block's points (direct global atomics when the image has too many tiles for LDS)
  for (int p = p0 + threadIdx.x; p < p1; p += 256) {
    const long long index = (long long)n * num_points + p;
    Footprint f;
    if (!footprint(coords, sigma, index, num_points, height, width, f) || !binned(f)) continue;
    const int tx0 = f.l / TS, tx1 = f.r / TS, ty0 = f.t / TS, ty1 = f.b / TS;
    for (int ty = ty0; ty <= ty1; ++ty)
      for (int tx = tx0; tx <= tx1; ++tx) {
        const int tile = ty * tiles_x + tx;
        if (lds) {
          atomicAdd(&s_hist[tile], 1);
        } else {
          const int slot = atomicAdd(gcount + tile, 1);
          if (FILL) lists[offsets[(size_t)n * tiles + tile] + slot] = (int)index;
        }
      }
  }
  if (!lds) return;
  __syncthreads();
  // one global atomic per tile this block touched; FILL: it returns the block's first slot in that tile's list
  for (int t = threadIdx.x; t < tiles; t += 256) {
    const int c = s_hist[t];
    if (c > 0) {
      const int first = atomicAdd(gcount + t, c);
      if (FILL) s_base[t] = offsets[(size_t)n * tiles + t] + first;
    }
    if (FILL) s_hist[t] = 0;               // re-used as the block-local cursor
  }
  if (!FILL) return;
  __syncthreads();
  for (int p = p0 + threadIdx.x; p < p1; p += 256) {
    const long long index = (long long)n * num_points + p;
    Footprint f;
    if (!footprint(coords, sigma, index, num_points, height, width, f) || !binned(f)) continue;
    const int tx0 = f.l / TS, tx1 = f.r / TS, ty0 = f.t / TS, ty1 = f.b / TS;
    for (int ty = ty0; ty <= ty1; ++ty)
      for (int tx = tx0; tx <= tx1; ++tx) {
        const int tile = ty * tiles_x + tx;
        lists[s_base[tile] + atomicAdd(&s_hist[tile], 1)] = (int)index;
      }
  }
}

// exclusive scan of `n` counts into offsets[0 .. n] (one block; n is tiles x images: thousands)
__global__ __launch_bounds__(1024) void splat_scan_kernel(const int* __restrict__ counts, int* __restrict__ offsets,
                                                          int n) {
  __shared__ int warp_tot[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    const int v = i < n ? counts[i] : 0;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(incl, off, 64);
      if (lane >= off) incl += u;
    }
    if (lane == 63) warp_tot[wid] = incl;
    __syncthreads();
    int before = carry;
    for (int w = 0; w < wid; ++w) before += warp_tot[w];
    if (i < n) offsets[i] = before + incl - v;
    __syncthreads();
    if (tid == 1023) carry = before + incl;
    __syncthreads();
  }
  if (tid == 0) offsets[n] = carry;
}

// ascending compare-exchange network (bitonic with mirrored first step): every exchange puts the minimum at the lower
// position, so entries beyond `len` act as +infinity without being stored
template <typename Get, typename Swap>
__device__ __forceinline__ void sort_network(int len, int tid, int nthreads, Get get, Swap swap) {
  int p2 = 1;
  while (p2 < len) p2 <<= 1;
  for (int k = 2; k <= p2; k <<= 1) {
    const int half = k >> 1;
    for (int i = tid; i < p2 / 2; i += nthreads) {
      const int blk = i / half, off = i - blk * half;
      const int lo = blk * k + off, hi = blk * k + k - 1 - off;
      if (hi < len && get(lo) > get(hi)) swap(lo, hi);
    }
    __syncthreads();
    for (int j = half >> 1; j >= 1; j >>= 1) {
      for (int i = tid; i < p2 / 2; i += nthreads) {
        const int lo = 2 * j * (i / j) + (i % j), hi = lo + j;
        if (hi < len && get(lo) > get(hi)) swap(lo, hi);
      }
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(256) void splat_sort_kernel(int* __restrict__ lists, const int* __restrict__ offsets) {
  __shared__ int s[kSortLds];
  const int tile = blockIdx.x;
  const int beg = offsets[tile], len = offsets[tile + 1] - beg;
  if (len <= 1) return;
  int* seg = lists + beg;
  if (len <= kSortLds) {
    for (int i = threadIdx.x; i < len; i += 256) s[i] = seg[i];
    __syncthreads();
    sort_network(len, threadIdx.x, 256, [&](int i) { return s[i]; },
                 [&](int a, int b) { const int t = s[a]; s[a] = s[b]; s[b] = t; });
    for (int i = threadIdx.x; i < len; i += 256) seg[i] = s[i];
  } else {        // a tile that holds more than 8192 points: the same network on the global segment
    sort_network(len, threadIdx.x, 256, [&](int i) { return seg[i]; },
                 [&](int a, int b) { const int t = seg[a]; seg[a] = seg[b]; seg[b] = t; });
  }
}

// ---- gather: one block per tile, thread = 4 consecutive pixels of one row --------------------------------------
// FUSED: out = (out + splat) / (max?(alpha_in + alpha, 1) + 1e-8) (splat_gpu.c:33-40); else alpha += , out += .
// Every wave walks the tile's whole (sorted) list on its own, 64 points at a time: lane l computes the box of point l,
// then the wave visits the 64 points through v_readlane - the box of a point is SCALAR data, so "none of my 8 rows"
// is a scalar branch that costs six instructions and no LDS round trip (the first version staged the boxes in LDS and
// paid its latency once per point: a dependent ds_read -> readfirstlane -> branch chain).
template <bool FUSED>
__global__ __launch_bounds__(256) void splat_tile_kernel(
    const float* __restrict__ coords, const float* __restrict__ values, const float* __restrict__ sigma,
    const int* __restrict__ lists, const int* __restrict__ offsets, float* __restrict__ alpha_io,
    float* __restrict__ output, int num_points, int channels, int height, int width, int tiles_x, int tiles_y,
    int soft) {
  constexpr int CG = 4;                  // channels per pass over the list
  const int tile = blockIdx.x;
  const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
  const int py = ty * TS + (tid >> 3), px0 = tx * TS + (tid & 7) * 4;
  const int wrow0 = ty * TS + wid * 8;                       // this wave's rows: wrow0 .. wrow0 + 7
  const int beg = offsets[tile], len = offsets[tile + 1] - beg;
  const size_t hw = (size_t)height * width;
  const bool row_ok = py < height;
  float* a_img = alpha_io ? alpha_io + (size_t)n * hw : nullptr;
  float* o_img = output + (size_t)n * channels * hw;

  for (int c0 = 0; c0 < channels; c0 += CG) {
    const int nc = channels - c0 < CG ? channels - c0 : CG;
    float acc_a[4] = {0.f, 0.f, 0.f, 0.f};
    float acc_v[CG][4];
#pragma unroll
    for (int c = 0; c < CG; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc_v[c][e] = 0.f;
    // three batches in flight: the list indices of batch b + 2 and the coordinates / values of batch b + 1 are being
    // fetched while batch b is visited (two dependent memory round trips per batch, hidden instead of paid 61 times)
    const float stdev = sigma[n];
    auto fetch_idx = [&](int base) { return (base + lane < len) ? lists[beg + base + lane] : -1; };
    struct Raw { float xc, yc, v[CG]; };
    auto fetch_raw = [&](int idx) {
      Raw r;
      r.xc = r.yc = -1.f;                                     // outside the image: an empty box
#pragma unroll
      for (int c = 0; c < CG; ++c) r.v[c] = 0.f;
      if (idx >= 0) {
        r.xc = coords[2 * (size_t)idx];
        r.yc = coords[2 * (size_t)idx + 1];
        const float* val = values + (size_t)idx * channels + c0;
#pragma unroll
        for (int c = 0; c < CG; ++c) r.v[c] = c < nc ? val[c] : 0.f;
      }
      return r;
    };
    int idx1 = fetch_idx(64);
    Raw raw0 = fetch_raw(fetch_idx(0));
    for (int base = 0; base < len; base += 64) {
      const Raw raw1 = fetch_raw(idx1);
      const int idx2 = fetch_idx(base + 128);
      // lane -> one point of this batch of 64
      Footprint f;
      if (!footprint_xy(raw0.xc, raw0.yc, stdev, height, width, f)) { f.t = 1; f.b = 0; f.l = 1; f.r = 0; }
      float pv[CG];
#pragma unroll
      for (int c = 0; c < CG; ++c) pv[c] = raw0.v[c];
      raw0 = raw1;
      idx1 = idx2;
      // does any of the 64 boxes reach this wave's rows?  (most batches of most waves: no)
      const bool mine = f.b >= wrow0 && f.t <= wrow0 + 7;
      unsigned long long todo = __ballot(mine);
      while (todo) {
        const int i = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int t = __builtin_amdgcn_readlane(f.t, i), b = __builtin_amdgcn_readlane(f.b, i);
        const int l = __builtin_amdgcn_readlane(f.l, i), r = __builtin_amdgcn_readlane(f.r, i);
        if (py < t || py > b || px0 + 3 < l || px0 > r) continue;
        const float xc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, f.xc), i));
        const float yc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, f.yc), i));
        const float nz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, f.normalizer), i));
        // (the values travel with the box in the owning lane's registers: a load inside this loop put one memory
        // latency on every visited point - 0.3 us x ~2000 points per wave was the whole kernel)
        float v[CG];
#pragma unroll
        for (int c = 0; c < CG; ++c)
          v[c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pv[c]), i));
        const float fy = (float)py - yc;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int px = px0 + e;
          if (px < l || px > r) continue;
          const float fx = (float)px - xc;
          const float alpha = __expf(nz * (fx * fx + fy * fy));       // v_exp_f32 (argument <= 0; results below 1e-12 do not matter)
          acc_a[e] += alpha;
#pragma unroll
          for (int c = 0; c < CG; ++c) acc_v[c][e] += alpha * v[c];
        }
      }
    }
    if (row_ok) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int px = px0 + e;
        if (px >= width) continue;
        const size_t pix = (size_t)py * width + px;
        if (FUSED) {
          float a = acc_a[e] + (a_img ? a_img[pix] : 0.f);
          if (soft) a = fmaxf(a, 1.0f);
          const float den = a + 1e-8f;
          for (int c = 0; c < nc; ++c) {
            float* o = o_img + (size_t)(c0 + c) * hw + pix;
            *o = (*o + acc_v[c][e]) / den;
          }
        } else {
          if (c0 == 0 && len > 0) a_img[pix] += acc_a[e];
          if (len > 0)
            for (int c = 0; c < nc; ++c) o_img[(size_t)(c0 + c) * hw + pix] += acc_v[c][e];
        }
      }
    }
  }
}

// bins + sorted lists for `top_count` points in the stream's scratch; -> device pointers
int build_bins(const float* coords, const float* sigma, int num_points, int height, int width, int top_count,
               int n_images, hipStream_t st, int& tiles_x, int& tiles_y, const int*& offsets, const int*& lists) {
  tiles_x = (width + TS - 1) / TS;
  tiles_y = (height + TS - 1) / TS;
  const long long tiles = (long long)n_images * tiles_x * tiles_y;
  if (tiles >= (1LL << 30) || (long long)top_count * 4 >= (1LL << 31)) return gg::fail(-2, "splat2d: problem too large");
  const size_t ints = (size_t)tiles * 2 + 1 + 1 + (size_t)top_count * 4;
  int* base = reinterpret_cast<int*>(gg::scratch(st, ints * sizeof(int)));
  if (!base) return -3;
  int* counts = base;                        // [tiles]   (count pass; re-used as the fill pass's cursor)
  int* offs = base + tiles;                  // [tiles + 1]
  int* lst = offs + tiles + 2;               // [4 * top_count]
  hipError_t e = hipMemsetAsync(counts, 0, sizeof(int) * (size_t)tiles, st);
  if (e != hipSuccess) return gg::fail((int)e, "splat2d: memset failed: %s", hipGetErrorString(e));
  const dim3 grid((unsigned)((num_points + kPtsPerBlock - 1) / kPtsPerBlock), (unsigned)n_images);
  if (n_images > 65535) return gg::fail(-2, "splat2d: too many images");
  splat_bin_kernel<false><<<grid, 256, 0, st>>>(coords, sigma, counts, nullptr, nullptr, num_points, height, width,
                                                 top_count, tiles_x, tiles_y);
  splat_scan_kernel<<<1, 1024, 0, st>>>(counts, offs, (int)tiles);
  e = hipMemsetAsync(counts, 0, sizeof(int) * (size_t)tiles, st);
  if (e != hipSuccess) return gg::fail((int)e, "splat2d: memset failed: %s", hipGetErrorString(e));
  splat_bin_kernel<true><<<grid, 256, 0, st>>>(coords, sigma, counts, offs, lst, num_points, height, width, top_count,
                                                tiles_x, tiles_y);
  splat_sort_kernel<<<(unsigned)tiles, 256, 0, st>>>(lst, offs);
  offsets = offs;
  lists = lst;
  return gg::launch_status("splat2d bins");
}

}  // namespace

extern "C" int gg_splat_forward_f32(const float* coords, const float* values, const float* sigma,
                                    float* alpha_splats, float* output, int num_points, int channels, int height,
                                    int width, int top_count, void* stream) {
  if (top_count <= 0) return 0;
  if (!coords || !values || !sigma || !alpha_splats || !output || num_points <= 0 || height <= 0 || width <= 0)
    return gg::fail(-2, "splat_forward: bad arguments");
  hipStream_t st = gg::as_stream(stream);
  const int n_images = (top_count + num_points - 1) / num_points;
  splat_large_kernel<<<gg::stream_grid((long long)top_count * 16, 256), 256, 0, st>>>(
      coords, values, sigma, alpha_splats, output, num_points, channels, height, width, top_count);
  int tiles_x, tiles_y;
  const int *offsets, *lists;
  if (int rc = build_bins(coords, sigma, num_points, height, width, top_count, n_images, st, tiles_x, tiles_y, offsets,
                          lists))
    return rc;
  splat_tile_kernel<false><<<(unsigned)(n_images * tiles_x * tiles_y), 256, 0, st>>>(
      coords, values, sigma, lists, offsets, alpha_splats, output, num_points, channels, height, width, tiles_x,
      tiles_y, 0);
  return gg::launch_status("splat_forward");
}

// the reference's symbol (splat_gpu_impl.cuh:11-22): stream first, no status
extern "C" void SplatForwardGpu(void* stream, const float* bottom_coordinates, const float* bottom_values,
                                const float* bottom_sigma, float* top_alpha_splats, float* top_output,
                                const int num_points_, const int channels_, const int height_, const int width_,
                                const int top_count) {
  (void)gg_splat_forward_f32(bottom_coordinates, bottom_values, bottom_sigma, top_alpha_splats, top_output, num_points_,
                             channels_, height_, width_, top_count, stream);
}

extern "C" int gg_splat2d_f32(float* output, float* alpha_ws, const float* input, const float* coords,
                              const float* values, const float* sigma, int n, int num_points, int channels,
                              int height, int width, int soft_normalize, void* stream) {
  const long long hw = (long long)height * width;
  const long long total = (long long)n * channels * hw;
  if (total <= 0) return 0;
  if (!output || !alpha_ws || !input) return gg::fail(-2, "splat2d: null pointer");
  if (num_points > 0 && (!coords || !values || !sigma)) return gg::fail(-2, "splat2d: null pointer");
  hipStream_t st = gg::as_stream(stream);
  hipError_t e = hipMemcpyAsync(output, input, sizeof(float) * (size_t)total, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return gg::fail((int)e, "splat2d: copy failed: %s", hipGetErrorString(e));
  e = hipMemsetAsync(alpha_ws, 0, sizeof(float) * (size_t)n * hw, st);
  if (e != hipSuccess) return gg::fail((int)e, "splat2d: memset failed: %s", hipGetErrorString(e));
  const long long top = (long long)n * num_points;
  if (top >= (1LL << 29)) return gg::fail(-2, "splat2d: too many points");
  const int top_count = (int)top;
  int tiles_x = (width + TS - 1) / TS, tiles_y = (height + TS - 1) / TS;
  const int *offsets = nullptr, *lists = nullptr;
  if (top_count > 0) {
    // boxes beyond 33 pixels first (float atomics onto the copy of the input / the zeroed alpha plane) ...
    splat_large_kernel<<<gg::stream_grid((long long)top_count * 16, 256), 256, 0, st>>>(
        coords, values, sigma, alpha_ws, output, num_points, channels, height, width, top_count);
    if (int rc = build_bins(coords, sigma, num_points, height, width, top_count, n, st, tiles_x, tiles_y, offsets, lists))
      return rc;
  } else {          // no points: every list is empty (offsets all zero)
    int* offs = reinterpret_cast<int*>(gg::scratch(st, sizeof(int) * ((size_t)n * tiles_x * tiles_y + 1)));
    if (!offs) return -3;
    e = hipMemsetAsync(offs, 0, sizeof(int) * ((size_t)n * tiles_x * tiles_y + 1), st);
    if (e != hipSuccess) return gg::fail((int)e, "splat2d: memset failed: %s", hipGetErrorString(e));
    offsets = offs;
    lists = offs;
  }
  // ... then every tile gathers its points, adds what is already there and writes each pixel once, normalised
  splat_tile_kernel<true><<<(unsigned)(n * tiles_x * tiles_y), 256, 0, st>>>(
      coords, values, sigma, lists, offsets, alpha_ws, output, num_points, channels, height, width, tiles_x, tiles_y,
      soft_normalize);
  return gg::launch_status("splat2d");
}
