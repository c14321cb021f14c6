// Which blocks of a launch share a CU?  Every block records where its first wave runs (XCC_ID, and
// the SE / CU fields of HW_ID) and when it started; the blocks spin long enough that the chip is
// full before any of them ends.  Same resources as conv_mfma_kernel<3,2>: 256 threads, two
// blocks per CU (launch bounds), 49 KB of dynamic LDS.
//   hipcc --offload-arch=gfx950 -O3 tools/wg_placement.hip -o tools/wg_placement.bin && tools/wg_placement.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ __launch_bounds__(256, 2) void probe(unsigned* out, long long spin) {
  extern __shared__ char lds[];
  if (threadIdx.x == 0) {
    lds[0] = 1;
    const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID, all 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 4 + 0] = hw;
    out[blockIdx.x * 4 + 1] = xcc;
    out[blockIdx.x * 4 + 2] = static_cast<unsigned>(t);
    out[blockIdx.x * 4 + 3] = static_cast<unsigned>(t >> 32);
  }
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {
  }
}

int main() {
  const int blocks = 1536;
  unsigned* d = nullptr;
  hipMalloc(&d, blocks * 16);
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 49152, 0, d, 20000LL);   // 20000 ticks of the 100 MHz clock = 200 us
  hipDeviceSynchronize();
  std::vector<unsigned> h(blocks * 4);
  hipMemcpy(h.data(), d, blocks * 16, hipMemcpyDeviceToHost);
  // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
  std::map<unsigned, std::vector<int>> where;
  unsigned long long tmin = ~0ULL;
  for (int b = 0; b < blocks; ++b) tmin = std::min(tmin, (static_cast<unsigned long long>(h[b * 4 + 3]) << 32) | h[b * 4 + 2]);
  for (int b = 0; b < blocks; ++b) {
    const unsigned hw = h[b * 4], xcc = h[b * 4 + 1] & 15;
    const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    where[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
  }
  printf("%zu distinct (xcc, se, sh, cu) places for %d blocks\n", where.size(), blocks);
  int shown = 0;
  for (const auto& kv : where) {
    if (shown++ >= 12) break;
    printf("xcc %u se %u sh %u cu %2u:", kv.first >> 12, (kv.first >> 8) & 15, (kv.first >> 4) & 15, kv.first & 15);
    for (int b : kv.second) {
      const unsigned long long t = ((static_cast<unsigned long long>(h[b * 4 + 3]) << 32) | h[b * 4 + 2]) - tmin;
      printf("  b%-4d(xi %-3d t %llu)", b, b >> 3, t / 100);
    }
    printf("\n");
  }
  // distance statistics between the first two blocks of a place, in xi = b >> 3
  std::map<int, int> dist;
  for (const auto& kv : where) {
    if (kv.second.size() >= 2) ++dist[(kv.second[1] >> 3) - (kv.second[0] >> 3)];
  }
  printf("xi distance between the first two blocks of a CU:");
  for (const auto& kv : dist) printf("  %d: %d CUs", kv.first, kv.second);
  printf("\n");
  return 0;
}
