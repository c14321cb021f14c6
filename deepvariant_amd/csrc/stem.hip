// stem.hip -- the Inception-v3 stem as two fused, persistent gfx950 kernels.
//
// Replaces the first five layers of tf_keras InceptionV3 as call_variants runs them
// (deepvariant/keras_modeling.py:268-274, deepvariant/call_variants.py:904-932;
// SURVEY.md App. B):
//     conv 3x3/2 C->32, conv 3x3 32->32            stem_a_kernel
//     conv 3x3 'same' 32->64, maxpool 3x3/2, conv 1x1 64->80     stem_b_kernel
// Per layer these five cost 2.0 MB of HBM traffic per candidate and ran at 4-5 TB/s,
// i.e. HBM-bound (profiles/r01q_op_trace.txt).  Fused, the tensors between them never leave
// the CU: a workgroup owns an image-aligned 2-D tile, keeps the producer's output tile
// (with the halo the consumer needs, recomputed per tile) in LDS, and only the 32-channel
// 47x108 and the 80-channel 23x53 tensors are written (0.52 MB per candidate).
//
// Structure shared by both kernels
//   * WEIGHTS LIVE IN REGISTERS.  A 3x3 32->32 filter bank is 18 MFMA A-fragments
//     (72 VGPRs per lane); every wave loads its fragments once and then walks over
//     many tiles (persistent grid), so the main loops carry no weight traffic at all:
//     one ds_read_b128 (the pixel fragment) per v_mfma_f32_32x32x16_f16.
//   * ACTIVATIONS LIVE IN LDS in the C8 piece layout [8-channel group][pixel][8]:
//     an MFMA B fragment (8 channels of one pixel) is one 16-byte piece, the 32 pixels
//     of a fragment are consecutive pieces of a tile row (conflict-free ds_read_b128),
//     and a filter tap is an immediate offset on the lane's base address.
//   * An accumulator leaves the matrix core with lane = pixel and registers = output
//     channels {0-3, 8-11, 16-19, 24-27} + 4*(lane>>5).  The LDS-resident intermediates
//     are written exactly like that (two 16-byte pieces per lane, no cross-lane
//     exchange) and the CONSUMER's weights are packed with the matching channel
//     permutation inside each 16-channel K chunk.
//   * BatchNorm shift: the accumulators start at the shift instead of zero.
//   * The next tile's global loads are issued before the current tile's last matrix
//     phase and written to LDS after it (before that phase's stores are issued, so no
//     s_waitcnt ever waits for a store); two workgroup barriers per tile.
#include <algorithm>
#include <type_traits>

#include "stem_fused.h"

namespace dv {
namespace {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
typedef unsigned uint3_t __attribute__((ext_vector_type(3)));

__device__ __forceinline__ half8_t lds_piece(const char* smem, unsigned off) {
  return *reinterpret_cast<const half8_t*>(smem + off);
}

// accumulator start value = folded BatchNorm shift, from the LDS table written in register
// order ([hi][16 registers])
__device__ __forceinline__ float16_t acc_init(const float* tbl) {
  const float4_t* t = reinterpret_cast<const float4_t*>(tbl);
  const float4_t a = t[0], b = t[1], c = t[2], d = t[3];
  return float16_t{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3],
                   c[0], c[1], c[2], c[3], d[0], d[1], d[2], d[3]};
}

// ReLU + fp16: registers 8*half .. 8*half+7 of an accumulator -> one 16-byte piece
__device__ __forceinline__ uint4_t relu_piece(const float16_t& a, int half) {
  const half2_t zero2 = {static_cast<_Float16>(0.f), static_cast<_Float16>(0.f)};
  uint4_t o;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float2_t v = {a[8 * half + 2 * k], a[8 * half + 2 * k + 1]};
    half2_t h = __builtin_convertvector(v, half2_t);
    h = __builtin_elementwise_max(h, zero2);
    o[k] = __builtin_bit_cast(unsigned, h);
  }
  return o;
}

// ReLU + fp16 + the lane-pair exchange that turns an accumulator into standard C8 pieces:
// returns piece[t] = output channels 8*(2t + hi) .. +7 of this lane's pixel (t = 0, 1).
__device__ __forceinline__ void relu_std_pieces(const float16_t& a, uint4_t (&piece)[2]) {
  const uint4_t lo = relu_piece(a, 0), up = relu_piece(a, 1);
  // lo = {q0.x, q0.y, q1.x, q1.y}, up = {q2.x, q2.y, q3.x, q3.y}; q = 4-cout quad 8q + 4hi
  const unsigned pk[4][2] = {{lo[0], lo[1]}, {lo[2], lo[3]}, {up[0], up[1]}, {up[2], up[3]}};
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const auto d0 = __builtin_amdgcn_permlane32_swap(pk[2 * t][0], pk[2 * t + 1][0], false, false);
    const auto d1 = __builtin_amdgcn_permlane32_swap(pk[2 * t][1], pk[2 * t + 1][1], false, false);
    piece[t] = uint4_t{d0[0], d1[0], d0[1], d1[1]};
  }
}

// S K-steps over NF pixel fragments with the weights in registers: the LDS pieces of step
// s + D are requested right after the MFMAs of step s, so D steps of matrix work cover the
// LDS latency.  Everything is unrolled (static ring slots); the sched_barriers pin the
// request order, which hipcc otherwise collapses into read -> wait -> MFMA per step.
template <int S, int NF, int D, class Addr>
__device__ __forceinline__ void mfma_sweep(const half8_t (&a)[S], const char* smem, Addr addr,
                                           float16_t (&acc)[NF]) {
  half8_t ring[D][NF];
#pragma unroll
  for (int d = 0; d < D && d < S; ++d) {
#pragma unroll
    for (int f = 0; f < NF; ++f) ring[d][f] = lds_piece(smem, addr(d, f));
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < S; ++s) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], ring[s % D][f], acc[f], 0, 0, 0);
    }
    if (s + D < S) {
#pragma unroll
      for (int f = 0; f < NF; ++f) ring[s % D][f] = lds_piece(smem, addr(s + D, f));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The same sweep for TWO output-channel subtiles that share every pixel piece (one LDS read
// per two MFMAs).
template <int S, int D, class Addr, class Filler>
__device__ __forceinline__ void mfma_sweep_pair(const half8_t (&a0)[S], const half8_t (&a1)[S],
                                                const char* smem, Addr addr, float16_t& acc0,
                                                float16_t& acc1, Filler filler) {
  half8_t ring[D];
#pragma unroll
  for (int d = 0; d < D && d < S; ++d) ring[d] = lds_piece(smem, addr(d));
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < S; ++s) {
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[s], ring[s % D], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[s], ring[s % D], acc1, 0, 0, 0);
    if (s + D < S) ring[s % D] = lds_piece(smem, addr(s + D));
    filler(s);  // a slice of unrelated work (DMA issue) that rides under the MFMAs
    __builtin_amdgcn_sched_barrier(0);
  }
}

// =========================================================================== stem A
constexpr int A_TH = kStemA_TH, A_TW = kStemA_TW;
constexpr int A_C1H = A_TH + 2, A_C1W = A_TW + 2;             // conv1 tile 9 x 56
constexpr int A_C1PX = A_C1H * A_C1W;                         // 504
constexpr int A_C1FR = (A_C1PX + 31) / 32;                    // 16 fragments
constexpr int A_C1PLANE = A_C1FR * 32 * 16;                   // bytes per 8-channel plane
constexpr int A_INH = 2 * A_C1H + 1, A_INW = 2 * A_C1W + 1;   // input patch 19 x 113
constexpr int A_INWH = (A_INW + 1) / 2;                       // 57 columns per parity plane
constexpr int A_INROW = A_INWH * 16;                          // 912 bytes
constexpr int A_INPLANE = A_INH * A_INROW;                    // 17328
constexpr int A_C2PX = A_TH * A_TW;                           // 378
constexpr int A_C2FR = (A_C2PX + 31) / 32;                    // 12 fragments
constexpr int A_OFF_C1 = 2 * A_INPLANE;                       // LDS map: input | conv1 | shifts
constexpr int A_OFF_SH = A_OFF_C1 + 4 * A_C1PLANE;
constexpr int A_LDS = A_OFF_SH + 64 * 4;
constexpr int A_THREADS = 256;
constexpr int A_ROWS_PER_PASS = A_THREADS / A_INW;            // 2
constexpr int A_PASSES = (A_INH + A_ROWS_PER_PASS - 1) / A_ROWS_PER_PASS;  // 10
static_assert(A_C1FR % 4 == 0 && A_C2FR % 4 == 0, "fragments split evenly over 4 waves");
static_assert(A_LDS <= 80 * 1024, "two workgroups per CU");
// Inputs of 9..12 channels (the long-read models: ONT_R104 9, PACBIO 10), round 6.  A pixel is up to 12 bytes: conv1's
// K chunk is ONE tap x 16 channels (k-group 0 = bytes 0..7, k-group 1 = bytes 8..15 of the pixel; bytes past C belong
// to the next pixel and meet zero weights), nine chunks.  The input patch stays uint8 in LDS -- four planes (column
// parity x k-group) of 8 bytes per pixel, the same 34,656 bytes as the two fp16 planes above, so the rest of the LDS
// map is shared -- and is normalised on its way into the matrix core (one v_perm + packed FMA per two channels, beside
// the MFMAs); conv1's nine weight fragments live in LDS (9 KB) instead of registers (36 VGPRs would not fit next to
// conv2's 72).  conv1's output tile, conv2 and the stores are the code of the <= 8-channel kernel.
constexpr int AW_ROW = A_INWH * 8;                            // 456 bytes: one row of one plane
constexpr int AW_PLANE = A_INH * AW_ROW;                      // 8664
static_assert(4 * AW_PLANE == 2 * A_INPLANE, "the uint8 patch of the wide kernel fills the fp16 patch's space");
constexpr int AW_OFF_W1 = A_OFF_SH + 64 * 4;
constexpr int AW_LDS = AW_OFF_W1 + kStemA_W1WideHalfs * 2;
static_assert(AW_LDS <= 80 * 1024, "two workgroups per CU");

// uint8 -> fp16 (x - 128) / 128 for 8 consecutive bytes: 0x6400 | b is the fp16 number
// 1024 + b, and (1024 + b) * 2^-7 - 9 = (b - 128) / 128 exactly (deepvariant/dv_utils.py:343-366)
__device__ __forceinline__ uint4_t normalise8(unsigned lo, unsigned up) {
  const half2_t scale = {static_cast<_Float16>(0.0078125f), static_cast<_Float16>(0.0078125f)};
  const half2_t bias = {static_cast<_Float16>(-9.0f), static_cast<_Float16>(-9.0f)};
  const unsigned k = 0x64646464u;
  const half2_t h01 = __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(lo, k, 0x00050004u));
  const half2_t h23 = __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(lo, k, 0x00070006u));
  const half2_t h45 = __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(up, k, 0x00050004u));
  const half2_t h67 = __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(up, k, 0x00070006u));
  return uint4_t{__builtin_bit_cast(unsigned, h01 * scale + bias),
                 __builtin_bit_cast(unsigned, h23 * scale + bias),
                 __builtin_bit_cast(unsigned, h45 * scale + bias),
                 __builtin_bit_cast(unsigned, h67 * scale + bias)};
}

template <bool WIDE>
__global__ __launch_bounds__(A_THREADS, 2) void stem_a_kernel(StemAArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;

  // ---- weights -> registers, once per wave ---------------------------------------------
  half8_t a1[5], a2[18];
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    if constexpr (!WIDE) a1[c] = *reinterpret_cast<const half8_t*>(p.w1 + ((c * 2 + hi) * 32 + l31) * 8);
  }
  if constexpr (WIDE) {   // conv1's nine (tap x 16 channels) fragments -> LDS, once per workgroup
    const uint4_t* src = reinterpret_cast<const uint4_t*>(p.w1);
    uint4_t* dst = reinterpret_cast<uint4_t*>(smem + AW_OFF_W1);
    for (int i = tid; i < kStemA_W1WideHalfs / 8; i += A_THREADS) dst[i] = src[i];
  }
#pragma unroll
  for (int kc = 0; kc < 18; ++kc) {
    a2[kc] = *reinterpret_cast<const half8_t*>(p.w2 + ((kc * 2 + hi) * 32 + l31) * 8);
  }
  // The weight loads retire HERE: inside the tile loop the only vector-memory traffic left
  // for s_waitcnt to reason about is the patch prefetch and the output stores.
  if constexpr (!WIDE) {
#pragma unroll
    for (int c = 0; c < 5; ++c) asm volatile("" : "+v"(a1[c]));
  }
#pragma unroll
  for (int kc = 0; kc < 18; ++kc) asm volatile("" : "+v"(a2[kc]));
  float* lsh = reinterpret_cast<float*>(smem + A_OFF_SH);
  if (tid < 64) {  // [conv][hi][register] -> output channel (r&3) + 8(r>>2) + 4hi
    const int r = tid & 15, h = (tid >> 4) & 1;
    lsh[tid] = (tid >= 32 ? p.shift2 : p.shift1)[(r & 3) + 8 * (r >> 2) + 4 * h];
  }

  // ---- per-thread constants of the three phases ----------------------------------------
  // staging: thread -> (row within the pass, column) of the input patch
  const int srow = tid >= A_INW ? 1 : 0;
  const int scol = tid - srow * A_INW;
  const bool sact = tid < A_ROWS_PER_PASS * A_INW;
  const unsigned srel = static_cast<unsigned>((srow * p.W + scol) * p.C);
  const unsigned spass = static_cast<unsigned>(A_ROWS_PER_PASS * p.W * p.C);
  const unsigned sdst = WIDE ? static_cast<unsigned>((scol & 1) * 2 * AW_PLANE + srow * AW_ROW + (scol >> 1) * 8)
                             : static_cast<unsigned>((scol & 1) * A_INPLANE + srow * A_INROW + (scol >> 1) * 16);
  // conv1: 4 fragments of the 9 x 56 tile per wave
  unsigned bA[4], bB[4], bC[4], c1dst[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int j = (wave * 4 + f) * 32 + l31;
    const int jj = min(j, A_C1PX - 1);
    const int cy = jj / A_C1W, cx = jj - cy * A_C1W;
    const unsigned base = static_cast<unsigned>(2 * cy * A_INROW + cx * 16);
    bA[f] = base + hi * A_INPLANE;  // tap pairs (kh,0),(kh,1): even / odd column plane
    bB[f] = base + hi * A_INROW;    // tap pair (0,2),(1,2): one row down
    bC[f] = base;                   // tap (2,2) + the zero-weight pad
    if constexpr (WIDE) {           // plane (column parity 0, k-group hi), row 2 cy, piece cx: taps are offsets on it
      bA[f] = static_cast<unsigned>(hi * AW_PLANE + 2 * cy * AW_ROW + cx * 8);
    }
    c1dst[f] = static_cast<unsigned>(A_OFF_C1 + hi * A_C1PLANE + j * 16);
  }
  // conv2: 3 fragments of the 7 x 54 tile per wave
  unsigned b2[3];
  int ty2[3], tx2[3];
  bool v2[3];
#pragma unroll
  for (int f = 0; f < 3; ++f) {
    const int i = (wave * 3 + f) * 32 + l31;
    v2[f] = i < A_C2PX;
    const int ii = min(i, A_C2PX - 1);
    ty2[f] = ii / A_TW;
    tx2[f] = ii - ty2[f] * A_TW;
    b2[f] = static_cast<unsigned>(A_OFF_C1 + hi * A_C1PLANE + (ty2[f] * A_C1W + tx2[f]) * 16);
  }

  const uint8_t* in_base = p.in_ind != nullptr ? *p.in_ind + p.in_off : p.in;   // wave-uniform
  {
    const unsigned long long v = reinterpret_cast<unsigned long long>(in_base);
    const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
    const unsigned up = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
    in_base = reinterpret_cast<const uint8_t*>((static_cast<unsigned long long>(up) << 32) | lo);
  }
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(in_base), 0, p.in_bytes, 0x00020000);
  const int tiles_img = p.tiles_y * p.tiles_x;
  const unsigned img_in = static_cast<unsigned>(p.H * p.W * p.C);
  const size_t img_out = static_cast<size_t>(p.og.groups) * p.og.hp * p.og.wp;  // pieces
  const unsigned gstride = static_cast<unsigned>(p.og.hp * p.og.wp);

  typedef typename std::conditional<WIDE, uint4_t, uint3_t>::type pre_t;
  pre_t pre[A_PASSES];
  auto tile_origin = [&](int t, int& n, int& y0, int& x0) {
    n = t / tiles_img;
    const int r = t - n * tiles_img;
    const int ty = r / p.tiles_x;
    y0 = ty * A_TH;
    x0 = (r - ty * p.tiles_x) * A_TW;
    return static_cast<unsigned>(n) * img_in + static_cast<unsigned>((2 * y0 * p.W + 2 * x0) * p.C);
  };
  auto issue_loads = [&](unsigned origin) {
#pragma unroll
    for (int ps = 0; ps < A_PASSES; ++ps) {
      const bool act = sact && (ps * A_ROWS_PER_PASS + srow < A_INH);
      const unsigned a = origin + srel + ps * spass;
      if constexpr (WIDE) {   // the aligned 16 bytes around the pixel's <= 12
        pre[ps] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, act ? (a & ~3u) : 0x80000000u, 0, 0);
      } else {
        pre[ps] = __builtin_amdgcn_raw_buffer_load_b96(rsrc, act ? (a & ~3u) : 0x80000000u, 0, 0);
      }
    }
  };

  // input patch: uint8 HWC -> fp16 pieces, even / odd columns in separate planes so that the
  // stride-2 taps of consecutive output pixels are consecutive pieces
  auto convert_patch = [&](unsigned org) {
#pragma unroll
    for (int ps = 0; ps < A_PASSES; ++ps) {
      if (sact && (ps * A_ROWS_PER_PASS + srow < A_INH)) {
        const unsigned a = org + srel + ps * spass;
        const unsigned s8 = (a & 3u) * 8u;
        const unsigned lo = __builtin_amdgcn_alignbit(pre[ps][1], pre[ps][0], s8);
        const unsigned up = __builtin_amdgcn_alignbit(pre[ps][2], pre[ps][1], s8);
        if constexpr (WIDE) {   // raw bytes: k-group 0 = bytes 0..7, k-group 1 = bytes 8..11 (+ 4 that meet zero weights)
          typedef unsigned uint2_t __attribute__((ext_vector_type(2)));
          const unsigned lo1 = __builtin_amdgcn_alignbit(pre[ps][3], pre[ps][2], s8);
          char* d = smem + sdst + ps * (A_ROWS_PER_PASS * AW_ROW);
          *reinterpret_cast<uint2_t*>(d) = uint2_t{lo, up};
          *reinterpret_cast<uint2_t*>(d + AW_PLANE) = uint2_t{lo1, 0u};
        } else {
          *reinterpret_cast<uint4_t*>(smem + sdst + ps * (A_ROWS_PER_PASS * A_INROW)) =
              normalise8(lo, up);
        }
      }
    }
  };
  // Blank-row skipping: the tile sequence of this workgroup is walked by next_tile(), which copies the
  // blank-determined tiles it passes (all 256 threads, global -> global, no LDS, no barrier) and stops at the
  // next tile that has to be computed.  Everything it decides on is workgroup-uniform.
  auto copy_blank_tile = [&](int nb_, int yb, int xb) {
    const uint4_t* src = reinterpret_cast<const uint4_t*>(p.blank_src);
    uint4_t* dst = reinterpret_cast<uint4_t*>(p.out) + static_cast<size_t>(nb_) * img_out;
    const int th = min(A_TH, p.OH2 - yb), tw = min(A_TW, p.OW2 - xb);
    const int per_group = th * tw;
    constexpr int kPer = (4 * A_TH * A_TW + A_THREADS - 1) / A_THREADS;   // 6 pieces per thread: all loads, then all stores
    uint4_t v[kPer];
    unsigned at[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int e = k * A_THREADS + tid;
      const int g = e / per_group, r = e - g * per_group;
      const int yy = r / tw, xx = r - yy * tw;
      at[k] = e < 4 * per_group ? static_cast<unsigned>(g) * gstride +
                                      static_cast<unsigned>((yb + yy + p.og.halo) * p.og.wp + xb + xx + p.og.halo)
                                : 0xffffffffu;
      if (at[k] != 0xffffffffu) v[k] = src[at[k]];
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      if (at[k] != 0xffffffffu) dst[at[k]] = v[k];
    }
  };
  auto next_tile = [&](int tt) {
    if (p.blank_thr == nullptr) return tt;
    while (tt < p.total_tiles) {
      int nb_, yb, xb;
      tile_origin(tt, nb_, yb, xb);
      if (yb < p.blank_thr[nb_]) break;
      if (yb < p.blank_need[nb_]) copy_blank_tile(nb_, yb, xb);   // (else: nobody reads this tile)
      tt += gridDim.x;
    }
    return tt;
  };
  int t = next_tile(blockIdx.x);
  int n, y0, x0;
  unsigned origin = 0;
  if (t < p.total_tiles) {
    origin = tile_origin(t, n, y0, x0);
    issue_loads(origin);
    convert_patch(origin);
  }
  while (t < p.total_tiles) {
    __syncthreads();  // patch complete; every wave is done with the previous conv1 tile

    // ---- conv1 (3x3 stride 2, K = 5 chunks of 2 taps x 8 channels) -> LDS ---------------
#pragma unroll
    for (int round = 0; round < 2; ++round) {
      float16_t acc[2];
#pragma unroll
      for (int f = 0; f < 2; ++f) acc[f] = acc_init(lsh + hi * 16);
      if constexpr (WIDE) {
        // nine taps x 16 channels: the A fragment of a tap from LDS (shared by the round's two pixel fragments), the
        // B fragments = 8 raw bytes of (row 2cy + kh, column 2cx + kw), normalised on the way in; the reads of tap
        // t + 1 are issued before the MFMAs of tap t
        typedef unsigned uint2_t __attribute__((ext_vector_type(2)));
        auto b_addr = [&](int tap, int f) {
          const int kh = tap / 3, kw = tap % 3;
          return bA[round * 2 + f] + static_cast<unsigned>((kw & 1) * 2 * AW_PLANE + kh * AW_ROW + (kw >> 1) * 8);
        };
        const char* w1l = smem + AW_OFF_W1 + (hi * 32 + l31) * 16;
        half8_t wa[2];
        uint2_t xb[2][2];
        wa[0] = lds_piece(w1l, 0);
#pragma unroll
        for (int f = 0; f < 2; ++f) xb[0][f] = *reinterpret_cast<const uint2_t*>(smem + b_addr(0, f));
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          if (tap + 1 < 9) {
            wa[(tap + 1) & 1] = lds_piece(w1l, (tap + 1) * 2 * 32 * 16);
#pragma unroll
            for (int f = 0; f < 2; ++f) xb[(tap + 1) & 1][f] = *reinterpret_cast<const uint2_t*>(smem + b_addr(tap + 1, f));
          }
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            const half8_t x = __builtin_bit_cast(half8_t, normalise8(xb[tap & 1][f][0], xb[tap & 1][f][1]));
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[tap & 1], x, acc[f], 0, 0, 0);
          }
        }
      } else
      mfma_sweep<5, 2, 3>(
          a1, smem,
          [&](int s, int f) {
            const int ff = round * 2 + f;
            return s < 3 ? bA[ff] + s * A_INROW : s == 3 ? bB[ff] + 16 : bC[ff] + 2 * A_INROW + 16;
          },
          acc);
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const int ff = round * 2 + f;
        // registers 0-7 = channels {0-3, 8-11} + 4hi = k-group hi of chunk 0; 8-15: chunk 1
        *reinterpret_cast<uint4_t*>(smem + c1dst[ff]) = relu_piece(acc[f], 0);
        *reinterpret_cast<uint4_t*>(smem + c1dst[ff] + 2 * A_C1PLANE) = relu_piece(acc[f], 1);
      }
    }
    __syncthreads();  // conv1 tile complete; the input patch may be overwritten

    // ---- next tile's pixels start their trip now, land after conv2's matrix work ----------
    const int tn = next_tile(t + gridDim.x);
    int nn = 0, yn = 0, xn = 0;
    unsigned origin_n = 0;
    if (tn < p.total_tiles) {
      origin_n = tile_origin(tn, nn, yn, xn);
      issue_loads(origin_n);
    }

    // ---- conv2 (3x3, 32 -> 32): 18 chunks, one LDS piece per MFMA -----------------------
    {
      float16_t acc[3];
#pragma unroll
      for (int f = 0; f < 3; ++f) acc[f] = acc_init(lsh + 32 + hi * 16);
      mfma_sweep<18, 3, 3>(
          a2, smem,
          [&](int s, int f) {
            const int tap = s >> 1, c = s & 1;
            return b2[f] + c * 2 * A_C1PLANE + ((tap / 3) * A_C1W + tap % 3) * 16;
          },
          acc);
      // The next patch goes to LDS BEFORE this tile's stores are issued: its s_waitcnt then
      // covers the prefetch loads only, and the stores get a whole tile to retire.
      if (tn < p.total_tiles) convert_patch(origin_n);
      uint4_t* outp = reinterpret_cast<uint4_t*>(p.out) + static_cast<size_t>(n) * img_out;
#pragma unroll
      for (int f = 0; f < 3; ++f) {
        uint4_t piece[2];
        relu_std_pieces(acc[f], piece);
        const int oy = y0 + ty2[f], ox = x0 + tx2[f];
        if (v2[f] && oy < p.OH2 && ox < p.OW2) {
          const unsigned o = static_cast<unsigned>((oy + p.og.halo) * p.og.wp + ox + p.og.halo);
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) outp[o + static_cast<unsigned>(2 * tt + hi) * gstride] = piece[tt];
        }
      }
    }
    t = tn;
    n = nn;
    y0 = yn;
    x0 = xn;
    origin = origin_n;
  }
}

// =========================================================================== stem B
constexpr int B_PH = kStemB_PH, B_PW = kStemB_PW;             // pooled tile 6 x 9 (wide build: 12 x 9)
constexpr int B_WAVES = kStemB_Waves;
constexpr int B_C3H = 2 * B_PH + 1, B_C3W = 2 * B_PW + 1;     // conv3 tile 13 x 19 (25 x 19)
constexpr int B_C3PX = B_C3H * B_C3W;                         // 247 (475)
constexpr int B_C3FR = (B_C3PX + 31) / 32;                    // 8 fragments (15)
// conv3 tile in LDS: 8 planes of 8 channels; the plane stride is an ODD number of pieces so
// that the max-pool's stride-2 reads of two adjacent planes interleave on the LDS banks.  Only
// the tile's own pixels are stored (the last fragment's surplus lanes are predicated off), so
// the stride is the pixel count rounded up to odd.
constexpr int B_C3PLANE = (B_C3PX | 1) * 16;                  // 3952 (7600)
constexpr int B_PTH = B_C3H + 2, B_PTW = B_C3W + 2;           // input patch 15 x 21 (27 x 21)
constexpr int B_PTPX = B_PTH * B_PTW;                         // 315 (567)
constexpr int B_PTPLANE = B_PTPX * 16;                        // 5040 (9072)
constexpr int B_PT_BYTES = (4 * B_PTPLANE + 1023) / 1024 * 1024;  // 32 channels, padded to the 1 KB DMA granule
constexpr int B_PT_PIECES = 4 * B_PTPX;                       // 1260 (2268)
constexpr int B_PPX = B_PH * B_PW;                            // 54 (108) pooled pixels
constexpr int B_PFR = (B_PPX + 31) / 32;                      // 2 (4) fragments
constexpr int B_PPLANE = B_PFR * 32 * 16;                     // 1024 (2048)
constexpr int B_THREADS = B_WAVES * 64;
constexpr int B_PASSES = (B_PT_PIECES + B_THREADS - 1) / B_THREADS;  // 5
constexpr int B_POOL_ROUNDS = (8 * B_PPX + B_THREADS - 1) / B_THREADS;  // 2
constexpr int B_OFF_C3 = 2 * B_PT_BYTES;                      // LDS map: patch x2 | conv3 | pooled | shifts
constexpr int B_OFF_P = (B_OFF_C3 + 8 * B_C3PLANE + 15) / 16 * 16;
constexpr int B_OFF_SH = B_OFF_P + 8 * B_PPLANE;
constexpr int B_LDS = B_OFF_SH + (64 + 96) * 4;
constexpr int B_BLOCKS_PER_CU = B_WAVES == 4 ? 2 : 1;
static_assert(B_LDS * B_BLOCKS_PER_CU <= 160 * 1024, "workgroups per CU");
static_assert(2 * B_PFR == B_WAVES, "1x1 work split: (pooled fragment, subtiles {0,1}) and (fragment, subtile 2)");
static_assert(B_C3FR <= 2 * B_WAVES, "two conv3 fragments per wave");

__global__ __launch_bounds__(B_THREADS, 2) void stem_b_kernel(StemBArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  // ---- conv3 weights -> registers: BOTH 32-channel halves (144 VGPRs), so that a pixel
  // piece read from LDS feeds two MFMAs ---------------------------------------------------
  half8_t a3lo[18], a3up[18];
#pragma unroll
  for (int kc = 0; kc < 18; ++kc) {
    a3lo[kc] = *reinterpret_cast<const half8_t*>(p.w3 + (((0 * 18 + kc) * 2 + hi) * 32 + l31) * 8);
    a3up[kc] = *reinterpret_cast<const half8_t*>(p.w3 + (((1 * 18 + kc) * 2 + hi) * 32 + l31) * 8);
  }
  // 1x1: the first half of the waves own output subtiles 0 and 1 of pooled fragment `wave`, the
  // second half subtile 2; its weights (12 fragments, L2 resident) are fetched per tile, not held
  const int s0 = wave < B_PFR ? 0 : 2;
  const int nsub = wave < B_PFR ? 2 : 1;
#pragma unroll
  for (int kc = 0; kc < 18; ++kc) {  // loads retire here (see stem A)
    asm volatile("" : "+v"(a3lo[kc]));
    asm volatile("" : "+v"(a3up[kc]));
  }
  float* lsh = reinterpret_cast<float*>(smem + B_OFF_SH);
  if (tid < 160) {  // [0,64): conv3 [cout half][hi][r]; [64,160): 1x1 [subtile][hi][r]
    const int r = tid & 15, h = (tid >> 4) & 1;
    const int co = (r & 3) + 8 * (r >> 2) + 4 * h;
    if (tid < 64) {
      lsh[tid] = p.shift3[(tid >> 5) * 32 + co];
    } else {
      const int c4 = ((tid - 64) >> 5) * 32 + co;
      lsh[tid] = c4 < p.Cout4 ? p.shift4[c4] : 0.f;
    }
  }

  // ---- per-thread constants ----------------------------------------------------------------
  unsigned prel[B_PASSES];   // patch piece -> byte offset relative to the tile origin
#pragma unroll
  for (int ps = 0; ps < B_PASSES; ++ps) {
    const int e = ps * B_THREADS + tid;
    const int g = e / B_PTPX, rem = e - g * B_PTPX;
    const int row = rem / B_PTW, col = rem - row * B_PTW;
    prel[ps] = e < B_PT_PIECES
                   ? static_cast<unsigned>(((g * p.ig.hp + row) * p.ig.wp + col) * 16)
                   : 0x80000000u;
  }
  // conv3: fragments wave and wave + B_WAVES of the tile
  unsigned b3[2], c3dst[2];
  bool c3ok[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int i = (wave + B_WAVES * m) * 32 + l31;
    c3ok[m] = i < B_C3PX;
    const int ii = min(i, B_C3PX - 1);
    const int cy = ii / B_C3W, cx = ii - cy * B_C3W;
    b3[m] = static_cast<unsigned>(hi * B_PTPLANE + (cy * B_PTW + cx) * 16);
    // conv3 output channel 32h + 16rh + {0-3, 8-11} + 4hi  ->  plane (2h + rh)*2 + hi
    c3dst[m] = static_cast<unsigned>(B_OFF_C3 + hi * B_C3PLANE + i * 16);
  }
  // 1x1: pooled fragment wave % B_PFR
  const int pp = (wave % B_PFR) * 32 + l31;
  const int ppc = min(pp, B_PPX - 1);
  const int ppy = ppc / B_PW, ppx = ppc - ppy * B_PW;
  const unsigned b4 = static_cast<unsigned>(B_OFF_P + hi * B_PPLANE + pp * 16);

  const int tiles_img = p.tiles_y * p.tiles_x;
  const unsigned ogstride = static_cast<unsigned>(p.og.hp * p.og.wp);

  auto tile_coords = [&](int t, int& n, int& py0, int& px0) {
    n = t / tiles_img;
    const int r = t - n * tiles_img;
    const int ty = r / p.tiles_x;
    py0 = ty * B_PH;
    px0 = (r - ty * p.tiles_x) * B_PW;
  };
  // The patch goes HBM -> LDS by LDS-DMA (no VGPRs: the 144 weight registers leave none to
  // spare): wave w moves the 1 KB granules w, w+8, ...; lane = piece within the granule.
  auto patch_desc = [&](int n) {
    const size_t off = static_cast<size_t>(n) * p.in_img_bytes;
    const size_t left = p.in_bytes - off;
    const unsigned long long base = reinterpret_cast<unsigned long long>(p.in) + off;
    // provably wave-uniform descriptor (no waterfall loop around the DMA)
    const unsigned blo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(base));
    const unsigned bup = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(base >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>((static_cast<unsigned long long>(bup) << 32) | blo), 0,
        static_cast<unsigned>(left < 0x7fffffffu ? left : 0x7fffffffu), 0x00020000);
  };
  // conv3 is 'same': output (oy, ox) reads rows oy-1..oy+1 = padded rows oy-1+halo..
  auto patch_soff = [&](int py0, int px0) {
    return static_cast<unsigned>(((2 * py0 - 1 + p.ig.halo) * p.ig.wp + 2 * px0 - 1 + p.ig.halo) * 16);
  };
  auto dma_round = [&](int ps, const __amdgpu_buffer_rsrc_t rsrc, unsigned soff, unsigned buf_off) {
    const unsigned g0 = static_cast<unsigned>((ps * B_WAVES + wave_u) * 1024);
    if (g0 < static_cast<unsigned>(B_PT_BYTES)) {  // wave-uniform
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsrc, (__attribute__((address_space(3))) void*)(smem + buf_off + g0), 16, prel[ps], soff, 0, 0);
    }
  };

  // Blank-row skipping (see stem A): next_tile() copies the blank-determined tiles it passes and stops at the next
  // one that has to be computed.
  auto copy_blank_tile = [&](int nb_, int pyb, int pxb) {
    const uint4_t* src = reinterpret_cast<const uint4_t*>(p.blank_src);
    uint4_t* dst = reinterpret_cast<uint4_t*>(p.out) +
                   static_cast<size_t>(nb_) * p.og.groups * ogstride;
    const int th = min(B_PH, p.PH - pyb), tw = min(B_PW, p.PW - pxb);
    const int per_group = th * tw;
    const int groups = (p.Cout4 + 7) / 8;
    constexpr int kPer = (12 * B_PH * B_PW + B_THREADS - 1) / B_THREADS;   // up to 96 couts: all loads, then all stores
    uint4_t v[kPer];
    unsigned at[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int e = k * B_THREADS + tid;
      const int g = e / per_group, r = e - g * per_group;
      const int yy = r / tw, xx = r - yy * tw;
      at[k] = e < groups * per_group ? static_cast<unsigned>(g) * ogstride +
                                           static_cast<unsigned>((pyb + yy + p.og.halo) * p.og.wp + pxb + xx + p.og.halo)
                                     : 0xffffffffu;
      if (at[k] != 0xffffffffu) v[k] = src[at[k]];
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      if (at[k] != 0xffffffffu) dst[at[k]] = v[k];
    }
  };
  auto next_tile = [&](int tt) {
    if (p.blank_thr == nullptr) return tt;
    while (tt < p.total_tiles) {
      int nb_, pyb, pxb;
      tile_coords(tt, nb_, pyb, pxb);
      if (pyb < p.blank_thr[nb_]) break;
      if (pyb < p.blank_need[nb_]) copy_blank_tile(nb_, pyb, pxb);   // (else: nobody reads this tile)
      tt += gridDim.x;
    }
    return tt;
  };
  int t = next_tile(blockIdx.x);
  int n = 0, py0 = 0, px0 = 0;
  unsigned buf = 0;  // byte offset of the current patch buffer
  if (t < p.total_tiles) {
    tile_coords(t, n, py0, px0);
    const __amdgpu_buffer_rsrc_t r0 = patch_desc(n);
    const unsigned so0 = patch_soff(py0, px0);
#pragma unroll
    for (int ps = 0; ps < B_PASSES; ++ps) dma_round(ps, r0, so0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  unsigned long long ph[7] = {0, 0, 0, 0, 0, 0, 0};
  unsigned long long tm = p.prof ? __builtin_amdgcn_s_memtime() : 0;
#define DV_PHASE(i_)                                          \
  if (p.prof) {                                               \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    ph[i_] += now_ - tm;                                      \
    tm = now_;                                                \
  }
  while (t < p.total_tiles) {
    const int tn = next_tile(t + gridDim.x);
    int nn = 0, pyn = 0, pxn = 0;
    const bool more = tn < p.total_tiles;
    if (more) tile_coords(tn, nn, pyn, pxn);
    // Next patch: one DMA per three sub-steps of the first fragment's sweep (a DMA's issue
    // slot then falls under the SIMD partner's MFMAs), waited for at barrier A.  Measured
    // alternatives: all five at the top of the tile +0.3 k cycles per tile; right after
    // barrier A of the previous tile (a whole tile of lead, HBM wait 1.4 k -> 0.3 k cycles)
    // +1.2 k cycles per tile -- the issue slots then sit in the MFMA-free pool phase.
    const __amdgpu_buffer_rsrc_t rn = patch_desc(more ? nn : n);
    const unsigned son = patch_soff(pyn, pxn);
    DV_PHASE(0)

    // ---- conv3 (3x3 'same', 32 -> 64) on this wave's two fragments -> LDS ---------------------
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      if (wave + B_WAVES * m < B_C3FR) {  // wave-uniform (wide build: fragment 15 does not exist)
        float16_t acc_lo = acc_init(lsh + hi * 16), acc_up = acc_init(lsh + 32 + hi * 16);
        mfma_sweep_pair<18, 4>(
            a3lo, a3up, smem,
            [&](int s) {
              const int tap = s >> 1, c = s & 1;
              return buf + b3[m] + c * 2 * B_PTPLANE + ((tap / 3) * B_PTW + tap % 3) * 16;
            },
            acc_lo, acc_up,
            [&](int sI) {
              if (m == 0 && more && sI % 3 == 0 && sI / 3 < B_PASSES) {
                dma_round(sI / 3, rn, son, buf ^ B_PT_BYTES);
              }
            });
        if (c3ok[m]) {   // the last fragment's surplus lanes have no slot in the tile
          *reinterpret_cast<uint4_t*>(smem + c3dst[m]) = relu_piece(acc_lo, 0);
          *reinterpret_cast<uint4_t*>(smem + c3dst[m] + 2 * B_C3PLANE) = relu_piece(acc_lo, 1);
          *reinterpret_cast<uint4_t*>(smem + c3dst[m] + 4 * B_C3PLANE) = relu_piece(acc_up, 0);
          *reinterpret_cast<uint4_t*>(smem + c3dst[m] + 6 * B_C3PLANE) = relu_piece(acc_up, 1);
        }
      }
    }
    // this wave's 1x1 weight fragments start their trip now (used after the pool)
    half8_t a4[8];
#pragma unroll
    for (int sI = 0; sI < 2; ++sI) {
      const int sub = min(s0 + sI, 2);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        a4[sI * 4 + c] = *reinterpret_cast<const half8_t*>(p.w4 + (((sub * 4 + c) * 2 + hi) * 32 + l31) * 8);
      }
    }
    DV_PHASE(1)
    if (p.prof) {  // tuning aid: how much of barrier A is the DMA wait
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      DV_PHASE(6)
    }
    // conv3 tile (LDS writes) and the next patch (my DMAs) are complete: one statement, so that
    // no LDS access moves across it
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    DV_PHASE(2)

    // ---- max-pool 3x3 / 2 on the LDS tile (per 8-channel plane) -------------------------------
    // lane -> (plane parity = lane & 1, pooled pixel): with the odd plane stride a half-wave's
    // 16-byte reads fall on 32 consecutive even/odd LDS slots -- no bank conflicts (they cost
    // 2x on 124 KB of reads per tile when a wave walked one plane with stride 2)
#pragma unroll
    for (int k = 0; k < B_POOL_ROUNDS; ++k) {
      const int e = k * B_THREADS + tid;
      if (e < 8 * B_PPX) {
        const int j = e >> 1;
        const int q = 2 * (j / B_PPX) + (e & 1), pq = j % B_PPX;
        const int qy = pq / B_PW, qx = pq - qy * B_PW;
        const char* src = smem + B_OFF_C3 + q * B_C3PLANE + (2 * qy * B_C3W + 2 * qx) * 16;
        // all nine pieces in flight before the first max (hipcc otherwise serialises
        // read -> wait -> max: nine exposed LDS latencies per pooled pixel)
        half8_t win[9];
#pragma unroll
        for (int w9 = 0; w9 < 9; ++w9) {
          win[w9] = *reinterpret_cast<const half8_t*>(src + ((w9 / 3) * B_C3W + w9 % 3) * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
        const half8_t m01 = __builtin_elementwise_max(win[0], win[1]);
        const half8_t m23 = __builtin_elementwise_max(win[2], win[3]);
        const half8_t m45 = __builtin_elementwise_max(win[4], win[5]);
        const half8_t m67 = __builtin_elementwise_max(win[6], win[7]);
        const half8_t best = __builtin_elementwise_max(
            __builtin_elementwise_max(__builtin_elementwise_max(m01, m23), __builtin_elementwise_max(m45, m67)),
            win[8]);
        *reinterpret_cast<half8_t*>(smem + B_OFF_P + q * B_PPLANE + pq * 16) = best;
      }
    }
    DV_PHASE(3)
    __syncthreads();  // pooled tile complete
    DV_PHASE(4)

    // ---- conv 1x1 (64 -> 80) on the pooled tile -> HBM ----------------------------------------
    {
      float16_t acc[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) acc[s] = acc_init(lsh + 64 + min(s0 + s, 2) * 32 + hi * 16);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const half8_t x = lds_piece(smem, b4 + c * 2 * B_PPLANE);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a4[c], x, acc[0], 0, 0, 0);
        if (nsub == 2) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a4[4 + c], x, acc[1], 0, 0, 0);
      }
      const int gy = py0 + ppy, gx = px0 + ppx;
      const bool ok = pp < B_PPX && gy < p.PH && gx < p.PW;
      uint4_t* outp = reinterpret_cast<uint4_t*>(p.out);
      const unsigned o = static_cast<unsigned>(
          (n * p.og.groups * p.og.hp + gy + p.og.halo) * p.og.wp + gx + p.og.halo);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (s < nsub) {
          uint4_t piece[2];
          relu_std_pieces(acc[s], piece);
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            const int group = (s0 + s) * 4 + 2 * tt + hi;
            if (ok && group * 8 < p.Cout4) outp[o + static_cast<unsigned>(group) * ogstride] = piece[tt];
          }
        }
      }
    }
    DV_PHASE(5)
    t = tn;
    n = nn;
    py0 = pyn;
    px0 = pxn;
    buf ^= B_PT_BYTES;
  }
#undef DV_PHASE
  if (p.prof && lane == 0 && (wave == 0 || wave == B_WAVES - 1)) {
#pragma unroll
    for (int i = 0; i < 7; ++i) p.prof[(blockIdx.x * 2 + (wave == B_WAVES - 1)) * 8 + i] = ph[i];
    // where the workgroup ran: HW_ID (wave / SIMD / CU / SH / SE / TG slot) and XCC_ID
    p.prof[(blockIdx.x * 2 + (wave == B_WAVES - 1)) * 8 + 7] =
        static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11))) |
        (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 15u) << 32);
  }
}

int cu_count(int device) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 256;
  return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
}

}  // namespace

// Blank-row skipping makes a tile's cost depend on its row: workgroup b walks tiles b, b + grid, b + 2 grid, ..., so
// with grid and the tiles per image sharing a factor it would only ever see some of the tile rows (512 workgroups,
// 24 tiles per image: three of them) -- and the workgroups that drew the upper rows would finish last.  A grid
// coprime with the tiles per image gives every workgroup every (row, column) equally often; at most a few workgroups
// of the persistent grid are given up.
static int balanced_grid(int grid, int tiles_img, bool skipping) {
  if (!skipping) return grid;
  auto gcd = [](int a, int b) {
    while (b) {
      const int t = a % b;
      a = b;
      b = t;
    }
    return a;
  };
  while (grid > 1 && gcd(grid, tiles_img) != 1) --grid;
  return grid;
}

int stem_a_blocks(int device) { return 2 * cu_count(device); }
int stem_b_blocks(int device) { return B_BLOCKS_PER_CU * cu_count(device); }

void launch_stem_a(const StemAArgs& a, int blocks, hipStream_t stream) {
  static const bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_a_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, A_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_a_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, AW_LDS);
    return true;
  }();
  (void)attr;
  const int grid = balanced_grid(std::max(1, std::min(blocks, a.total_tiles)), a.tiles_y * a.tiles_x, a.blank_thr != nullptr);
  if (a.C > 8) {
    hipLaunchKernelGGL(stem_a_kernel<true>, dim3(grid), dim3(A_THREADS), AW_LDS, stream, a);
  } else {
    hipLaunchKernelGGL(stem_a_kernel<false>, dim3(grid), dim3(A_THREADS), A_LDS, stream, a);
  }
}

void launch_stem_b(const StemBArgs& a, int blocks, hipStream_t stream) {
  static const bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_b_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS);
    return true;
  }();
  (void)attr;
  const int grid = balanced_grid(std::max(1, std::min(blocks, a.total_tiles)), a.tiles_y * a.tiles_x, a.blank_thr != nullptr);
  hipLaunchKernelGGL(stem_b_kernel, dim3(grid), dim3(B_THREADS), B_LDS, stream, a);
}

// ---------------------------------------------------------------------------- weight packing
// All images are [chunk][k-group g][32 couts][8 k]: lane (cout = lane & 31, g = lane >> 5)
// reads its MFMA A fragment as one 16-byte piece.

// channel held at position j of k-group g of 16-channel chunk c when the PRODUCER wrote its
// accumulator registers straight to LDS (see the file header)
static int permuted_channel(int c, int g, int j) { return 16 * c + (j < 4 ? j : j + 4) + 4 * g; }

// conv1 tap of (chunk, k-group): pairs whose LDS addresses differ by a lane-half constant
static int stem_a_tap(int chunk, int g) {
  static const int taps[5][2] = {{0, 1}, {3, 4}, {6, 7}, {2, 5}, {8, -1}};
  return taps[chunk][g];
}

void pack_stem_a_w1(const float* w, const float* inv, int cin, _Float16* dst) {
  std::fill(dst, dst + kStemA_W1Halfs, static_cast<_Float16>(0.f));
  for (int c = 0; c < 5; ++c)
    for (int g = 0; g < 2; ++g) {
      const int tap = stem_a_tap(c, g);
      if (tap < 0) continue;
      for (int co = 0; co < 32; ++co)
        for (int ci = 0; ci < cin && ci < 8; ++ci) {
          dst[((c * 2 + g) * 32 + co) * 8 + ci] =
              static_cast<_Float16>(w[(static_cast<size_t>(tap) * cin + ci) * 32 + co] * inv[co]);
        }
    }
}

void pack_stem_a_w1_wide(const float* w, const float* inv, int cin, _Float16* dst) {
  std::fill(dst, dst + kStemA_W1WideHalfs, static_cast<_Float16>(0.f));
  for (int tap = 0; tap < 9; ++tap)
    for (int g = 0; g < 2; ++g)
      for (int co = 0; co < 32; ++co)
        for (int j = 0; j < 8; ++j) {
          const int ci = 8 * g + j;   // input channel = byte of the pixel
          if (ci >= cin) continue;
          dst[((tap * 2 + g) * 32 + co) * 8 + j] =
              static_cast<_Float16>(w[(static_cast<size_t>(tap) * cin + ci) * 32 + co] * inv[co]);
        }
}

void pack_stem_a_w2(const float* w, const float* inv, _Float16* dst) {
  for (int tap = 0; tap < 9; ++tap)
    for (int c = 0; c < 2; ++c)
      for (int g = 0; g < 2; ++g)
        for (int co = 0; co < 32; ++co)
          for (int j = 0; j < 8; ++j) {
            const int ci = permuted_channel(c, g, j);
            dst[(((tap * 2 + c) * 2 + g) * 32 + co) * 8 + j] =
                static_cast<_Float16>(w[(static_cast<size_t>(tap) * 32 + ci) * 32 + co] * inv[co]);
          }
}

void pack_stem_b_w3(const float* w, const float* inv, _Float16* dst) {
  for (int h = 0; h < 2; ++h)
    for (int tap = 0; tap < 9; ++tap)
      for (int c = 0; c < 2; ++c)
        for (int g = 0; g < 2; ++g)
          for (int co = 0; co < 32; ++co)
            for (int j = 0; j < 8; ++j) {
              const int ci = 16 * c + 8 * g + j;  // input comes from HBM in the standard order
              const int cout = 32 * h + co;
              dst[((((h * 18 + tap * 2 + c) * 2) + g) * 32 + co) * 8 + j] = static_cast<_Float16>(
                  w[(static_cast<size_t>(tap) * 32 + ci) * 64 + cout] * inv[cout]);
            }
}

void pack_stem_b_w4(const float* w, const float* inv, int cout_n, _Float16* dst) {
  std::fill(dst, dst + kStemB_W4Halfs, static_cast<_Float16>(0.f));
  for (int s = 0; s < 3; ++s)
    for (int c = 0; c < 4; ++c)
      for (int g = 0; g < 2; ++g)
        for (int co = 0; co < 32; ++co) {
          const int cout = 32 * s + co;
          if (cout >= cout_n) continue;
          for (int j = 0; j < 8; ++j) {
            const int ci = permuted_channel(c, g, j);
            dst[(((s * 4 + c) * 2 + g) * 32 + co) * 8 + j] =
                static_cast<_Float16>(w[static_cast<size_t>(ci) * cout_n + cout] * inv[cout]);
          }
        }
}

}  // namespace dv
