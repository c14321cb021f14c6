// L1 (vector cache) -> VGPR throughput microbenchmark for gfx950: which 16-byte-per-lane
// load shapes reach the 64 B/clk/CU the TA/TCP path is rated at?
//   hipcc --offload-arch=gfx950 -O3 tools/l1_bw.hip -o tools/l1_bw.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));

// MODE 0: global_load_dwordx4, wave reads 1 KB contiguous
// MODE 1: raw_buffer_load_b128, 1 KB contiguous
// MODE 2: raw_buffer_load_b128, two 512-byte runs 64 KB apart (our C8 fragment shape)
// MODE 3: raw_buffer_load_b128 with an SGPR soffset, two runs (exactly the conv kernel's form)
// MODE 4: global_load_dwordx2 (8 B/lane), contiguous 512 B
// MODE 5: ds_read_b128 from LDS (reference)
template <int MODE>
__global__ __launch_bounds__(256) void probe(const uint4_t* src, uint4_t* sink, int iters, unsigned bytes) {
  __shared__ uint4_t lds[4096];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[i];
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4_t*>(src), 0, bytes, 0x00020000);
  uint4_t acc = {0, 0, 0, 0};
  // each wave cycles over a 16 KB window (4 waves -> 64 KB... keep it L1-resident: 8 KB/wave)
  unsigned off = wave * 8192 + (MODE == 2 || MODE == 3 ? ((lane & 31) * 16 + (lane >> 5) * 4096) : lane * 16);
  if (MODE == 4) off = wave * 8192 + lane * 8;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned o = off + (u & 3) * 1024 * (MODE == 2 || MODE == 3 ? 0 : 1) + (MODE == 2 || MODE == 3 ? (u & 3) * 512 : 0);
      uint4_t v;
      if (MODE == 0) {
        v = *reinterpret_cast<const uint4_t*>(reinterpret_cast<const char*>(src) + o);
      } else if (MODE == 1 || MODE == 2) {
        v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, 0);
      } else if (MODE == 3) {
        v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, __builtin_amdgcn_readfirstlane((u & 3) * 512), 0);
      } else if (MODE == 4) {
        const uint2 t = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(src) + off + (u & 3) * 512);
        v = uint4_t{t.x, t.y, 0, 0};
      } else {
        v = lds[(wave * 512 + (u & 3) * 64 + lane) & 4095];
      }
      acc ^= v;
      asm volatile("" ::: "memory");
    }
  }
  if (acc[0] == 0x12345 && acc[1] == 0x777) sink[threadIdx.x] = acc;
}

template <int MODE>
void run(const char* name, const uint4_t* src, uint4_t* sink, unsigned bytes, double clk_ghz, int n_cu) {
  const int iters = 4000, blocks = n_cu * 4;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  probe<MODE><<<blocks, 256>>>(src, sink, 10, bytes);
  hipEventRecord(a);
  probe<MODE><<<blocks, 256>>>(src, sink, iters, bytes);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double per_lane = MODE == 4 ? 8 : 16;
  const double total = static_cast<double>(blocks) * 256 * iters * 8 * per_lane;
  printf("%-58s %8.2f TB/s  = %6.1f B/clk/CU at %.2f GHz\n", name, total / ms / 1e9,
         total / (ms * 1e-3) / n_cu / (clk_ghz * 1e9), clk_ghz);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const double ghz = p.clockRate / 1e6;
  const unsigned bytes = 1 << 20;
  uint4_t *src, *sink;
  hipMalloc(&src, bytes);
  hipMemset(src, 1, bytes);
  hipMalloc(&sink, 1 << 16);
  printf("CUs %d clock %.2f GHz, 4 blocks x 4 waves per CU\n", p.multiProcessorCount, ghz);
  run<0>("global_load_dwordx4, 1 KB contiguous per wave", src, sink, bytes, ghz, p.multiProcessorCount);
  run<1>("buffer_load_dwordx4 offen, 1 KB contiguous", src, sink, bytes, ghz, p.multiProcessorCount);
  run<2>("buffer_load_dwordx4 offen, 2 x 512 B runs (C8 fragment)", src, sink, bytes, ghz, p.multiProcessorCount);
  run<3>("buffer_load_dwordx4 offen + soffset, 2 x 512 B runs", src, sink, bytes, ghz, p.multiProcessorCount);
  run<4>("global_load_dwordx2, 512 B contiguous", src, sink, bytes, ghz, p.multiProcessorCount);
  run<5>("ds_read_b128 (LDS reference)", src, sink, bytes, ghz, p.multiProcessorCount);
  return 0;
}
