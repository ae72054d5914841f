// Does v_mfma_f32_32x32x16_{bf16,f16} stall when consecutive MFMAs of a wave alternate between only TWO accumulator
// tiles (the transposed-convolution tile: 6 MFMAs per unit on acc[cls][0..1]) instead of four (the stride-1 tile)?
// Build: hipcc -O3 --offload-arch=gfx950 mfma_dep_probe.hip -o mfma_dep_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)(float)(e + 1); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 12; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % NACC], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 1234.5f) out[0] = s;
}

template <int NACC>
void run(int threads, const char* what) {
  float* out;
  hipMalloc(&out, 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<256, threads>>>(out, 1000);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC><<<256, threads>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * 12 * (threads / 64) / 4.0;     // one block per CU
  printf("%-28s %d accumulators, %d waves/SIMD: %8.3f ms  -> %6.1f ns per MFMA slot per SIMD (32 cycles at 2.4 GHz = 13.3 ns)\n",
         what, NACC, threads / 256, ms, ms * 1e6 / mfma_per_simd);
  hipFree(out);
}

int main() {
  run<1>(256, "bf16 32x32x16"); run<2>(256, "bf16 32x32x16"); run<4>(256, "bf16 32x32x16");
  run<1>(512, "bf16 32x32x16"); run<2>(512, "bf16 32x32x16"); run<4>(512, "bf16 32x32x16");
  return 0;
}
