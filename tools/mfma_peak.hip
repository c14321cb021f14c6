// Micro-benchmark: what does v_mfma_f32_32x32x16_f16 sustain on this chip with the
// accumulator count / occupancy of conv_mfma_kernel?  (tuning aid, not product)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
  half8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i * 0.01f); }
  float16_t acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu, int iters) {
  float* out; hipMalloc(&out, 256 * 256 * 16 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  mfma_loop<NACC><<<grid, 256>>>(out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  mfma_loop<NACC><<<grid, 256>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * 32 * 32 * 16 * NACC * (double)iters * grid * 4;
  printf("NACC=%d blocks/CU=%d iters=%d: %.3f ms, %.1f TFLOP/s\n", NACC, blocks_per_cu, iters, ms, flops / ms / 1e9);
  hipFree(out);
}

int main() {
  run<1>(1, 20000); run<2>(1, 10000); run<4>(1, 5000); run<6>(1, 4000); run<6>(2, 4000); run<8>(1, 3000); run<4>(2, 5000); run<4>(4, 5000);
  // short-lived blocks: the same total work split in many small launches of small blocks
  run<6>(2, 270); run<6>(2, 540); run<6>(4, 270);
  return 0;
}
