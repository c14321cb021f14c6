// Do 16-byte-per-lane stores that do not cover whole 128-byte lines cost HBM reads (line fills)?
// Every wave instruction stores floor(64 / lanes) runs of `run` bytes (lanes = run / 16), run r at
// out + shift + r * pitch: dense aligned runs, runs shifted off the line grid, rows of a map with
// a halo (pitch > run).  Times them, and under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE   (and a second pass with WRITE_SIZE)
// the per-dispatch counters show whether partially written lines are filled from HBM first.
//   hipcc --offload-arch=gfx950 -O3 tools/store_align.hip -o /tmp/store_align && /tmp/store_align
#include <hip/hip_runtime.h>

#include <cstdio>

typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));

__global__ void store_runs(char* out, size_t n_runs, int shift, int lanes_per_run, int pitch) {
  const size_t wave = (blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const size_t n_waves = (static_cast<size_t>(gridDim.x) * blockDim.x) >> 6;
  const int runs_per_instr = 64 / lanes_per_run;
  const int sub = lane / lanes_per_run, pos = lane - sub * lanes_per_run;
  for (size_t r = wave * runs_per_instr; r < n_runs; r += n_waves * runs_per_instr) {
    const size_t run = r + sub;
    if (sub < runs_per_instr && run < n_runs) {
      *reinterpret_cast<uint4_t*>(out + shift + run * static_cast<size_t>(pitch) + pos * 16) =
          uint4_t{1u, 2u, 3u, static_cast<unsigned>(run)};
    }
  }
}

int main() {
  const size_t bytes = size_t(3) << 30;
  char* d = nullptr;
  if (hipMalloc(&d, bytes + 4096) != hipSuccess) return 1;
  (void)hipMemset(d, 0, bytes + 4096);
  struct Case {
    const char* name;
    int shift, run, pitch;
  } cases[] = {{"512 B runs, dense, aligned", 0, 512, 512},
               {"512 B runs, dense, shifted 48 B", 48, 512, 512},
               {"192 B runs, dense (4x12 rows, no halo)", 0, 192, 192},
               {"192 B runs at 288 B pitch (4x12 rows, halo 3)", 48, 192, 288},
               {"400 B runs, dense (10x25 rows, no halo)", 0, 400, 400},
               {"400 B runs at 432 B pitch (10x25 rows, halo 1)", 16, 400, 432},
               {"816 B runs, dense (21x51 rows)", 0, 816, 816},
               {"128 B runs at 256 B pitch, aligned", 0, 128, 256},
               {"64 B runs at 128 B pitch (half lines)", 0, 64, 128},
               {"64 B runs at 128 B pitch, shifted 32 B", 32, 64, 128}};
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (const Case& c : cases) {
    const size_t n_runs = (bytes - 4096) / c.pitch;
    float ms_total = 0;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      store_runs<<<4096, 256>>>(d, n_runs, c.shift, c.run / 16, c.pitch);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep) ms_total += ms;
    }
    const double gb = static_cast<double>(n_runs) * c.run / 1e9;
    printf("%-48s %7.3f ms  %6.2f GB stored  %7.1f GB/s\n", c.name, ms_total / 2, gb, gb / (ms_total / 2 * 1e-3));
  }
  return 0;
}
