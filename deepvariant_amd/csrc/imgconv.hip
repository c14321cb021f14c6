// imgconv.hip -- stride-1 convolutions of the Inception blocks over tiles of WHOLE feature
// maps, both MFMA operands in LDS.
//
// The Inception-v3 blocks of call_variants' classifier (deepvariant/keras_modeling.py:268-274,
// SURVEY.md App. B) work on small maps (10x25, 4x12, 1x5 at the WGS pileup shape), so a tile
// of G consecutive images (G*OH*OW <= 512 output pixels = 16 MFMA fragments) is contiguous
// in the C8 layout and needs no halo exchange.  Compared with conv_mfma_kernel (model.hip),
// which pulls every pixel fragment of every filter tap through the vector L1:
//   * the input patch of a 16-channel chunk (G images x (OH+KH-1) x (OW+KW-1) pixels) is
//     copied into LDS ONCE by LDS-DMA (global_load_lds_dwordx4, per-lane gather addresses,
//     no VGPRs) and all KH*KW taps read it from there: 9x / 25x / 7x fewer L1 requests;
//   * the weight slab of the step goes through LDS the same way;
//   * workgroups are persistent and the (tile, K-step) sequence is ONE software pipeline:
//     the DMA of step i+1 -- the next tile's first step included -- is in flight while step
//     i multiplies; one s_barrier per step, no wave ever waits on a register load.
// A step = KC channel chunks x all taps (KC = 1 for filters with taps, 4 for 1x1).
#include <cstdlib>

#include "imgconv.h"

namespace dv {
namespace {

using namespace convk;

constexpr int IC_WAVES = 8;
constexpr int IC_PT = 2;                       // pixel fragments per wave
constexpr int IC_THREADS = 64 * IC_WAVES;
constexpr int IC_MAXA = 8;                     // activation DMA rounds per step (8 KB each)

typedef __attribute__((address_space(3))) void* lptr_t;

// s_waitcnt needs an immediate: at most `n` of this wave's vector-memory operations may still
// be in flight (they complete in issue order), then the workgroup barrier -- one statement, so
// that neither LDS reads nor later DMAs can move across it.
__device__ __forceinline__ void wait_dma_and_barrier(int n) {
#define DV_CASE(k_) case k_: asm volatile("s_waitcnt vmcnt(" #k_ ") lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
  switch (n) {
    DV_CASE(1) DV_CASE(2) DV_CASE(3) DV_CASE(4) DV_CASE(5) DV_CASE(6) DV_CASE(7) DV_CASE(8)
    DV_CASE(9) DV_CASE(10) DV_CASE(11) DV_CASE(12) DV_CASE(13) DV_CASE(14) DV_CASE(15) DV_CASE(16)
    default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
  }
#undef DV_CASE
}

// R = LDS ring depth: the DMA of step i + R - 1 is issued while step i multiplies.  R = 2
// gives a DMA one step (~3.5 k cycles) of lead, less than an HBM round trip under load
// (measured ~5 k cycles): every step then waits.  R = 3 is used wherever three slabs fit.
template <int KH, int KW, int KC, int NB, int R>
__global__ __launch_bounds__(IC_THREADS, 2) void imgconv_kernel(ImgConvArgs p) {
  constexpr int TAPS = KH * KW, S = KC * TAPS, BN = NB * 32, PT = IC_PT;
  constexpr int W_SLAB = S * 2 * BN * 16;      // bytes: [kc][tap][k-group][BN couts][8 halfs]
  constexpr int W_ROUNDS = (W_SLAB / 1024 + IC_WAVES - 1) / IC_WAVES;
  constexpr int D = 2;                         // LDS fragment prefetch depth (steps of NB*PT MFMAs)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned act_bytes = static_cast<unsigned>(p.act_slab_bytes);
  const unsigned w_lds0 = R * act_bytes;

  // ---- per-lane constants ---------------------------------------------------------------
  const ConvArgs& c = p.c;
  const int patch_px = p.RP * p.CP;
  unsigned bbase[PT];
  int pimg[PT], poh[PT], pow_[PT];
  bool pvalid[PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int m = (wave * PT + pt) * 32 + l31;
    pvalid[pt] = m < p.G * p.P;
    const int mm = pvalid[pt] ? m : 0;
    const int img = mm / p.P, pix = mm - img * p.P;
    const int oy = pix / c.OW, ox = pix - oy * c.OW;
    pimg[pt] = img;
    poh[pt] = oy;
    pow_[pt] = ox;
    // slab layout per chunk: [k-group][image][patch row][patch col] pieces
    bbase[pt] = static_cast<unsigned>(((hi * p.G + img) * patch_px + oy * p.CP + ox) * 16);
  }
  const unsigned abase = static_cast<unsigned>((hi * BN + l31) * 16);
  const unsigned chunk_lds = static_cast<unsigned>(2 * p.plane_pieces * 16);
  const unsigned cp16 = static_cast<unsigned>(p.CP * 16);
  // activation DMA: piece (round k, this wave, lane) of the slab -> byte offset in the input
  // relative to (first image of the tile, first chunk of the step)
  unsigned arel[IC_MAXA];
#pragma unroll
  for (int k = 0; k < IC_MAXA; ++k) {
    const int e = (k * IC_WAVES + wave) * 64 + lane;
    unsigned off = 0;
    if (e < p.act_pieces) {
      const int kc = e / (2 * p.plane_pieces), r1 = e - kc * 2 * p.plane_pieces;
      const int kg = r1 / p.plane_pieces, r2 = r1 - kg * p.plane_pieces;
      const int img = r2 / patch_px, r3 = r2 - img * patch_px;
      const int r = r3 / p.CP, cc = r3 - r * p.CP;
      off = static_cast<unsigned>(img) * c.img_bytes +
            static_cast<unsigned>((((2 * kc + kg) * c.ig.hp + r + c.ig.halo - c.pad_h) * c.ig.wp + cc +
                                   c.ig.halo - c.pad_w) * 16);
    }
    arel[k] = off;
  }

  const int total = p.n_img_tiles * p.n_cout_tiles;
  auto first_image = [&](int item) {
    const int t = item / p.n_cout_tiles;
    return max(0, min(t * p.G, c.N - p.G));  // the last tile is shifted back to end at image N
  };
  // Buffer-addressed DMA: descriptor (SGPRs) = tile / step base, voffset = the lane's 32-bit
  // gather offset -- no 64-bit per-lane addresses.
  // (hipcc keeps the running 64-bit step pointers in VGPRs and would wrap every DMA in a
  // waterfall loop: the readfirstlanes make the descriptor provably wave-uniform)
  auto uniform_ptr = [](const char* q) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(q);
    const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
    const unsigned up = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
    return reinterpret_cast<char*>((static_cast<unsigned long long>(up) << 32) | lo);
  };
  // The DMAs of one step, as a list of J = IC_MAXA + W_ROUNDS wave-instructions: issued one
  // or two at a time BETWEEN the sub-steps of the running step, so that a wave's DMA issue
  // slots (tens of cycles each) fall under its SIMD partner's MFMAs instead of after a barrier
  // where every wave of the CU would issue them at once.
  constexpr int J = IC_MAXA + W_ROUNDS;
  auto act_desc = [&](int n0, int st) {
    return __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(reinterpret_cast<const char*>(c.in) + static_cast<size_t>(n0) * c.img_bytes +
                    static_cast<size_t>(st) * KC * c.chunk_stride),
        0, 0x7fffffff, 0x00020000);
  };
  auto w_desc = [&](int ct, int st) {
    return __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(reinterpret_cast<const char*>(c.w) + (static_cast<size_t>(ct) * p.n_steps + st) * W_SLAB),
        0, W_SLAB, 0x00020000);
  };
  auto issue_one = [&](int j, const __amdgpu_buffer_rsrc_t ra, const __amdgpu_buffer_rsrc_t rw,
                       unsigned slot) {
    if (j < IC_MAXA) {
      const unsigned piece0 = static_cast<unsigned>((j * IC_WAVES + wave) * 1024);
      if (piece0 < act_bytes) {  // wave-uniform
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(smem + slot * act_bytes + piece0), 16,
                                                 arel[j < IC_MAXA ? j : 0], 0, 0, 0);
      }
    } else {
      const unsigned piece0 = static_cast<unsigned>(((j - IC_MAXA) * IC_WAVES + wave) * 1024);
      if (piece0 < W_SLAB) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(smem + w_lds0 + slot * W_SLAB + piece0), 16,
                                                 lane * 16, piece0, 0, 0);
      }
    }
  };

  // the (item, K step) sequence of this workgroup as ONE stream of steps
  struct Pos {
    int item, st, n0, ct;
  };
  auto advance = [&](Pos q) {
    if (q.st + 1 < p.n_steps) {
      ++q.st;
      return q;
    }
    q.item += gridDim.x;
    q.st = 0;
    if (q.item < total) {
      q.n0 = first_image(q.item);
      q.ct = q.item % p.n_cout_tiles;
    }
    return q;
  };
  auto issue_all = [&](const Pos& q, unsigned slot) {
    const __amdgpu_buffer_rsrc_t ra = act_desc(q.n0, q.st), rw = w_desc(q.ct, q.st);
#pragma unroll
    for (int j = 0; j < J; ++j) issue_one(j, ra, rw, slot);
  };
  int my_dmas = 0;  // DMA instructions this wave issues per step (wave-uniform)
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const unsigned piece0 = static_cast<unsigned>(((j < IC_MAXA ? j : j - IC_MAXA) * IC_WAVES + wave) * 1024);
    my_dmas += (j < IC_MAXA ? piece0 < act_bytes : piece0 < static_cast<unsigned>(W_SLAB)) ? 1 : 0;
  }

  my_dmas = __builtin_amdgcn_readfirstlane(my_dmas);

  Pos cur{static_cast<int>(blockIdx.x), 0, 0, 0};
  if (cur.item >= total) return;
  cur.n0 = first_image(cur.item);
  cur.ct = cur.item % p.n_cout_tiles;
  Pos ahead = cur;
  unsigned slot = 0;
#pragma unroll
  for (int k = 0; k < R - 1; ++k) {  // steps 0 .. R-2 start their trip before the loop
    if (ahead.item < total) issue_all(ahead, k);
    ahead = advance(ahead);
  }
  float16_t acc[NB][PT];
  for (;;) {
    if (cur.st == 0) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[nb][pt][i] = 0.f;
    }
    // this step's DMAs have landed (the R - 2 younger steps' may still fly), every wave is
    // past the LDS reads of the slot the next DMAs will overwrite
    if (R == 2) {
      wait_dma_and_barrier(0);
    } else {
      const Pos nx = advance(cur);
      wait_dma_and_barrier(nx.item < total ? my_dmas : 0);
    }
    const bool more = ahead.item < total;
    const __amdgpu_buffer_rsrc_t ra = act_desc(more ? ahead.n0 : cur.n0, more ? ahead.st : 0);
    const __amdgpu_buffer_rsrc_t rw = w_desc(more ? ahead.ct : cur.ct, more ? ahead.st : 0);
    const unsigned slot_ahead = (slot + R - 1) % R;
    // ---- S = KC * taps sub-steps of NB x PT MFMAs; fragments D sub-steps ahead -------------
    const char* aslab = smem + slot * act_bytes;
    const char* wslab = smem + w_lds0 + slot * W_SLAB + abase;
    half8_t wr[D][NB], xr[D][PT];
    auto load_sub = [&](int s, int d) {
      const int kc = s / TAPS, tap = s - kc * TAPS;
      const unsigned toff = kc * chunk_lds + (tap / KW) * cp16 + (tap % KW) * 16;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        wr[d][nb] = *reinterpret_cast<const half8_t*>(wslab + s * (2 * BN * 16) + nb * 512);
      }
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        xr[d][pt] = *reinterpret_cast<const half8_t*>(aslab + bbase[pt] + toff);
      }
    };
#pragma unroll
    for (int d = 0; d < D && d < S; ++d) load_sub(d, d);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < S; ++s) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          acc[nb][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[s % D][nb], xr[s % D][pt],
                                                               acc[nb][pt], 0, 0, 0);
        }
      if (s + D < S) load_sub(s + D, s % D);
      __builtin_amdgcn_sched_barrier(0);
      if (more) {  // this sub-step's share of the DMAs of step i + R - 1
#pragma unroll
        for (int j = s * J / S; j < (s + 1) * J / S; ++j) issue_one(j, ra, rw, slot_ahead);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (cur.st == p.n_steps - 1) {
      // ---- epilogue: shift + ReLU, 16-byte pieces into the branch tensors -------------------
      int pn[PT];
      bool mvalid[PT];
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        pn[pt] = cur.n0 + pimg[pt];
        mvalid[pt] = pvalid[pt] && pn[pt] < c.N;
      }
      conv_epilogue<NB, PT>(acc, c, cur.ct, pn, poh, pow_, mvalid, lane);
    }
    cur = advance(cur);
    if (cur.item >= total) break;
    ahead = advance(ahead);
    slot = (slot + 1) % R;
  }
}

template <int KH, int KW, int KC, int NB, int R>
void launch_ring(const ImgConvArgs& a, int blocks, size_t lds, hipStream_t stream) {
  static const bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(imgconv_kernel<KH, KW, KC, NB, R>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)attr;
  const int total = a.n_img_tiles * a.n_cout_tiles;
  const int grid = total < blocks ? total : blocks;
  hipLaunchKernelGGL((imgconv_kernel<KH, KW, KC, NB, R>), dim3(grid > 0 ? grid : 1), dim3(IC_THREADS), lds,
                     stream, a);
}

template <int KH, int KW, int KC, int NB>
void launch_one(const ImgConvArgs& a, int blocks, size_t /*lds2*/, hipStream_t stream) {
  // Two slabs in flight.  Three (DV_IMGCONV_RING=3, where the CU's LDS holds them) measured
  // 3-5 % SLOWER on every 10 x 25 layer: the step wait is not DMA latency.
  static const bool ring3 = getenv("DV_IMGCONV_RING") != nullptr && atoi(getenv("DV_IMGCONV_RING")) == 3;
  const size_t slab = static_cast<size_t>(a.act_slab_bytes) + imgconv_wslab_halfs(KH, KW, NB) * 2;
  if (ring3 && 3 * slab <= 160 * 1024) {
    launch_ring<KH, KW, KC, NB, 3>(a, blocks, 3 * slab, stream);
  } else {
    launch_ring<KH, KW, KC, NB, 2>(a, blocks, 2 * slab, stream);
  }
}

template <int KH, int KW, int KC>
void launch_nb(const ImgConvArgs& a, int nb, int blocks, size_t lds, hipStream_t stream) {
  switch (nb) {
    case 2: launch_one<KH, KW, KC, 2>(a, blocks, lds, stream); break;
    case 3: launch_one<KH, KW, KC, 3>(a, blocks, lds, stream); break;
    default: launch_one<KH, KW, KC, 4>(a, blocks, lds, stream); break;
  }
}

}  // namespace

int imgconv_threads() { return IC_THREADS; }

int imgconv_kc(int kh, int kw) { return kh * kw == 1 ? 4 : 1; }

bool imgconv_supported(int kh, int kw, int nb) {
  if (nb < 2 || nb > 4) return false;
  return (kh == 1 && kw == 1) || (kh == 3 && kw == 3) || (kh == 5 && kw == 5) ||
         (kh == 1 && kw == 7) || (kh == 7 && kw == 1) || (kh == 1 && kw == 3) || (kh == 3 && kw == 1);
}

size_t imgconv_wslab_halfs(int kh, int kw, int nb) {
  return static_cast<size_t>(imgconv_kc(kh, kw)) * kh * kw * 2 * nb * 32 * 8;
}

size_t imgconv_lds_bytes(const ImgConvArgs& a, int nb) {
  return 2 * (static_cast<size_t>(a.act_slab_bytes) + imgconv_wslab_halfs(a.c.KH, a.c.KW, nb) * 2);
}

void launch_imgconv(const ImgConvArgs& a, int nb, int blocks, hipStream_t stream) {
  const size_t lds = imgconv_lds_bytes(a, nb);
  const int kh = a.c.KH, kw = a.c.KW;
  if (kh == 1 && kw == 1) launch_nb<1, 1, 4>(a, nb, blocks, lds, stream);
  else if (kh == 3 && kw == 3) launch_nb<3, 3, 1>(a, nb, blocks, lds, stream);
  else if (kh == 5 && kw == 5) launch_nb<5, 5, 1>(a, nb, blocks, lds, stream);
  else if (kh == 1 && kw == 7) launch_nb<1, 7, 1>(a, nb, blocks, lds, stream);
  else if (kh == 7 && kw == 1) launch_nb<7, 1, 1>(a, nb, blocks, lds, stream);
  else if (kh == 1 && kw == 3) launch_nb<1, 3, 1>(a, nb, blocks, lds, stream);
  else launch_nb<3, 1, 1>(a, nb, blocks, lds, stream);
}

}  // namespace dv
