// Micro-benchmark (tuning aid): how many independent accumulators does ONE wave need to keep a SIMD's matrix
// pipe busy?  Cycles per MFMA (s_memtime) for NACC accumulators issued round-robin, with one and with two
// waves per SIMD, for v_mfma_f32_32x32x16_f16 and v_mfma_f32_16x16x32_f16.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_dep.hip -o /tmp/mfma_dep && /tmp/mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef float float4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void loop32(float* out, unsigned long long* cyc, int iters) {
  half8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i * 0.01f); }
  float16_t acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
__global__ __launch_bounds__(256) void loop16(float* out, unsigned long long* cyc, int iters) {
  half8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i * 0.01f); }
  float4_t acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int i = 0; i < 4; ++i) acc[n][i] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[n], 0, 0, 0);
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) for (int i = 0; i < 4; ++i) s += acc[n][i];
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class K>
void run(const char* name, K kernel, int nacc, int blocks_per_cu, int iters, double flop_per_mfma) {
  float* out; hipMalloc(&out, 256 * 256 * 16 * sizeof(float));
  unsigned long long* cyc; hipMalloc(&cyc, 4096 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  kernel<<<grid, 256>>>(out, cyc, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kernel<<<grid, 256>>>(out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[4096];
  hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (int i = 0; i < grid; ++i) mean += h[i]; mean /= grid;
  const double mfmas = (double)nacc * iters;
  printf("%s NACC=%d waves/SIMD=%d: %.3f ms, %.0f TFLOP/s, %.1f s_memtime ticks per MFMA per wave (%.1f per SIMD)\n", name, nacc,
         blocks_per_cu, ms, flop_per_mfma * mfmas * grid * 4 / ms / 1e9, mean / mfmas, mean / mfmas / blocks_per_cu);
  hipFree(out); hipFree(cyc);
}

int main() {
  const double f32 = 2.0 * 32 * 32 * 16, f16 = 2.0 * 16 * 16 * 32;
  run("32x32x16", loop32<1>, 1, 1, 8000, f32); run("32x32x16", loop32<2>, 2, 1, 4000, f32); run("32x32x16", loop32<3>, 3, 1, 3000, f32);
  run("32x32x16", loop32<4>, 4, 1, 2000, f32); run("32x32x16", loop32<1>, 1, 2, 8000, f32); run("32x32x16", loop32<2>, 2, 2, 4000, f32);
  run("32x32x16", loop32<4>, 4, 2, 2000, f32);
  run("16x16x32", loop16<1>, 1, 1, 16000, f16); run("16x16x32", loop16<2>, 2, 1, 8000, f16); run("16x16x32", loop16<4>, 4, 1, 4000, f16);
  run("16x16x32", loop16<8>, 8, 1, 2000, f16); run("16x16x32", loop16<2>, 2, 2, 8000, f16); run("16x16x32", loop16<4>, 4, 2, 4000, f16);
  run("16x16x32", loop16<8>, 8, 2, 2000, f16);
  return 0;
}
