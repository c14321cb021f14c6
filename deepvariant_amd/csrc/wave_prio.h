// Asymmetric wave priority between the workgroups that share a CU.
//
// The fused / tiled kernels of the classifier run two workgroups per CU so that one workgroup's MFMA-free
// phases (DMA wait, barriers, pooling, epilogue stores) can hide under the other's matrix work.  With equal
// priorities that only happens by luck: two workgroups that enter their MFMA phase together share the
// SIMD's matrix pipe at half speed each, leave it together, and then both sit in their MFMA-free phases with the
// pipe idle -- stem_b's phase profile (conv3 takes 4.1 K cycles for 2.3 K cycles of MFMA work)
// is exactly that.  Giving ONE of the two a higher instruction-issue priority (s_setprio) breaks the symmetry:
// the favoured workgroup runs its matrix phase at full speed, the other gets the pipe while the first is in
// its MFMA-free phases, and the two settle in anti-phase.  Results cannot change: priorities only reorder
// instruction issue between waves.
//
// Which of the two: the workgroup's slot on the CU (HW_ID.TG_ID, gfx9 layout: bits 19:16) -- co-resident
// workgroups hold different slots by construction.  DV_PRIO (bit mask over kernels, host side) and DV_PRIO_MODE
// (1 = TG_ID parity, 2 = wave-slot parity HW_ID.WAVE_ID, 3 = blockIdx / 256 parity for persistent grids of two
// workgroups per CU) are tuning knobs; the defaults are what measured best.
#ifndef DV_WAVE_PRIO_H_
#define DV_WAVE_PRIO_H_

#include <hip/hip_runtime.h>

#include <cstdlib>

namespace dv {

enum PrioKernel {
  kPrioStemB = 1,
  kPrioStemA = 2,
  kPrioConvMfma = 4,
  kPrioImgconv = 8,
  kPrioResident = 16,
  kPrioChain = 32,
};

constexpr int kPrioDefaultMask = 0;

// Host: the mode a kernel of this kind is launched with (0 = every wave at the default priority).
inline int prio_mode(int kernel_bit) {
  static const int mask = getenv("DV_PRIO") ? atoi(getenv("DV_PRIO")) : kPrioDefaultMask;
  static const int mode = getenv("DV_PRIO_MODE") ? atoi(getenv("DV_PRIO_MODE")) : 1;
  return (mask & kernel_bit) ? mode : 0;
}

// Device: called once at the top of a kernel; `mode` is a kernel argument (wave-uniform).
__device__ __forceinline__ void asym_priority(int mode) {
  if (mode == 0) return;
  unsigned key;
  if (mode == 1) {
    key = __builtin_amdgcn_s_getreg((4 << 0) | (16 << 6) | (3 << 11));   // HW_REG_HW_ID, TG_ID = bits 19:16
  } else if (mode == 2) {
    key = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (3 << 11));    // WAVE_ID = bits 3:0
  } else {
    key = blockIdx.x >> 8;
  }
  if (key & 1u) __builtin_amdgcn_s_setprio(3);
}

}  // namespace dv

#endif  // DV_WAVE_PRIO_H_
