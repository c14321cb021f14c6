// chain.hip -- the factorised-7x7 branches of Inception-v3's 17x17 blocks as fused,
// persistent gfx950 kernels: 1x7 -> 7x1 (-> 1x7 -> 7x1 -> 1x7) with every intermediate
// tensor resident in LDS.
//
// Replaces the per-layer launches of tf_keras InceptionV3's mixed4..mixed7 branch7x7 /
// branch7x7dbl chains and mixed8's branch7x7x3 head (deepvariant/keras_modeling.py:268-274
// builds the backbone; SURVEY.md App. B has the graph).  Per layer these launches spent about
// half of their time outside the K loop (prologue, output stores, fill / drain:
// profiles/r02_conv_ablation.txt) and fetched their pixel operand through the vector L1,
// which is what bounded them (TA 87-89 % busy, HISTORY.md 7).  Here
//   * a workgroup owns a tile of G WHOLE images (G * h * w <= 192 pixels = 6 MFMA fragments)
//     and walks the whole chain on it: the c-channel intermediate (<= 74 KB) lives in ONE LDS
//     buffer that is rewritten in place between layers -- a layer's full output sits in the
//     accumulators of the four computing waves when its input dies;
//   * the pixel operand of every MFMA is a ds_read_b128 of that buffer (tile pixels are stored
//     [map row][image][column], so a filter tap is a constant byte offset and a fragment of 32
//     pixels spans at most two map rows: the k x 1 layers skip the taps that only meet the
//     zero padding, like conv_mfma_kernel's row-band mode);
//   * WAVE SPECIALISATION: waves 0-3 (one per SIMD) only read LDS and issue MFMAs; waves 4-7
//     only move data -- the weight slab of the next 16-channel chunk (all taps, all output
//     channels: 28-43 KB) and the next tile's input, by LDS-DMA (buffer_load ... lds) -- and
//     meet the computing waves at one s_barrier per chunk.  The computing waves never wait on
//     a vector-memory counter except for their own output stores.
// Two tile shapes: the 17x17 stage's small maps (<= 96 pixels: G whole maps in a 192-pixel tile,
// waves = 2 pixel halves x 2 cout halves, 1 x k / k x 1 filters) and the 35x35 stage's maps (up to
// 256 pixels: one or two maps in a 256-pixel tile, waves = 4 pixel quarters x all <= 96 couts,
// 3x3 / 5x5 filters: the 3x3 -> 3x3 branch of mixed0..2 and the 5x5 layers).
// The K order (channel chunk major, tap minor), the fp32 accumulation, the fp16 rounding of
// every intermediate and the shift + ReLU are those of the per-layer kernels (model.hip), and
// skipped taps only ever multiply zeros: results are bit-identical to the per-layer path
// (tests/test_hip_chain.py).
#include <cstdlib>
#include <type_traits>

#include "chain.h"

namespace dv {
namespace {

using namespace convk;

constexpr int CH_THREADS = 512;

typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void barrier_after_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void barrier_after_dma() {
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void barrier_only() { asm volatile("s_barrier" ::: "memory"); }

__device__ __forceinline__ const char* uniform_ptr(const char* q) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(q);
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
  const unsigned up = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
  return reinterpret_cast<const char*>((static_cast<unsigned long long>(up) << 32) | lo);
}

// Which of a wave's NB x 3 (cout subtile, pixel fragment) MFMA tiles exist.  A layer of five
// subtiles (160 channels) cannot be halved: 3 + 2 subtiles leaves two SIMDs idle a third of the
// time.  Instead the pixel half's 15 tiles go 8 + 7: the first wave takes subtiles 0-2 WITHOUT
// (subtile 2, fragment 2), the second subtiles 2-4 with ONLY fragment 2 of subtile 2.
//   SKIP 0: all NB x 3     SKIP 1: without (nb 2, pt 2)     SKIP 2: without (nb 0, pt 0) and (nb 0, pt 1)
template <int SKIP>
__device__ __forceinline__ constexpr bool chain_tile(int nb, int pt) {
  return SKIP == 1 ? !(nb == 2 && pt == 2) : SKIP == 2 ? !(nb == 0 && pt < 2) : true;
}

// One 16-channel chunk: NT filter taps x the wave's MFMA tiles.  The fragments of tap i + 1 are
// requested between the MFMAs of tap i (two static register sets, one request per MFMA slot).
// PRE: the first tap's pixel fragments were requested by the caller BEFORE the chunk barrier (the
// activation tile does not change inside a layer; only the weight slab waits for the barrier).
// KW: 0 = one-dimensional filter, tap i sits i * b_tap_stride bytes from the first (b_tap_stride =
// one pixel or one tile row); > 0 = KW-wide two-dimensional filter, tap i = (i / KW) tile rows
// (b_tap_stride bytes each) + (i % KW) pixels.
template <int NB, int PT, int NT, int KW, int SKIP, bool PRE>
__device__ __forceinline__ void chain_step(const char* smem, unsigned a_addr, unsigned a_tap_stride,
                                           const unsigned (&b_addr)[PT], unsigned b_tap_stride,
                                           const unsigned (&mask)[PT], unsigned zero_addr,
                                           const half8_t (&b_pre)[PT], float16_t (&acc)[NB][PT]) {
  half8_t A[2][NB], B[2][PT];
  auto load_a = [&](int i, int s) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      A[s][nb] = *reinterpret_cast<const half8_t*>(smem + a_addr + i * a_tap_stride + nb * 512);
    }
  };
  auto load_b = [&](int i, int s) {
    const unsigned off = KW == 0 ? i * b_tap_stride : (i / (KW ? KW : 1)) * b_tap_stride + (i % (KW ? KW : 1)) * 16u;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const unsigned ad = (mask[pt] >> i) & 1u ? b_addr[pt] + off : zero_addr;
      B[s][pt] = *reinterpret_cast<const half8_t*>(smem + ad);
    }
  };
  auto load = [&](int i, int s) {
    load_a(i, s);
    load_b(i, s);
  };
  load_a(0, 0);
  if (PRE) {
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) B[0][pt] = b_pre[pt];
  } else {
    load_b(0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    if (i + 1 < NT) load(i + 1, (i + 1) & 1);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        if (chain_tile<SKIP>(nb, pt)) {
          acc[nb][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[i & 1][nb], B[i & 1][pt], acc[nb][pt], 0, 0, 0);
        }
      }
    }
    if (i + 1 < NT) {
      // the next tap's NB + PT requests (and their address selects) ride in the issue slots
      // between this tap's MFMAs, one request per MFMA, instead of in a gap after them
#pragma unroll
      for (int k = 0; k < NB + PT; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);     // MFMA
        __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);     // VALU
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

typedef float float4_t __attribute__((ext_vector_type(4)));

// shift + ReLU + fp16 of one 32-cout accumulator, lanes l / l+32 paired into standard C8 pieces:
// piece[t] = output channels cbase + 8 * (2t + hi) .. +7 of this lane's pixel (conv_common.h's
// epilogue arithmetic, so that values match the per-layer kernels bit for bit).  sh[q] = the
// shifts of couts cbase + 8q + 4hi .. +3, loaded by the caller (one 16-byte load per quad, all in
// flight together -- scalar loads here cost a round trip each: 7 k cycles per layer, measured).
__device__ __forceinline__ void chain_pieces(const float16_t& a, const float4_t (&sh)[4], uint4_t (&piece)[2]) {
  const half2_t zero2 = {static_cast<_Float16>(0.f), static_cast<_Float16>(0.f)};
  unsigned pk[4][2];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float2_t v0 = float2_t{a[4 * q], a[4 * q + 1]} + float2_t{sh[q][0], sh[q][1]};
    const float2_t v1 = float2_t{a[4 * q + 2], a[4 * q + 3]} + float2_t{sh[q][2], sh[q][3]};
    half2_t h0 = __builtin_convertvector(v0, half2_t), h1 = __builtin_convertvector(v1, half2_t);
    h0 = __builtin_elementwise_max(h0, zero2);
    h1 = __builtin_elementwise_max(h1, zero2);
    pk[q][0] = __builtin_bit_cast(unsigned, h0);
    pk[q][1] = __builtin_bit_cast(unsigned, h1);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const auto d0 = __builtin_amdgcn_permlane32_swap(pk[2 * t][0], pk[2 * t + 1][0], false, false);
    const auto d1 = __builtin_amdgcn_permlane32_swap(pk[2 * t][1], pk[2 * t + 1][1], false, false);
    piece[t] = uint4_t{d0[0], d1[0], d0[1], d1[1]};
  }
}

// ------------------------------------------------------------------ computing waves (0-3)
// PT = 3 (192-pixel tiles): wave = (pixel half ph, cout half ch): fragments 3 ph .. 3 ph + 2, the
//   first / second half of the layer's 32-cout subtiles (8 + 7 tiles for five subtiles, chain_tile).
// PT = 2 (256-pixel tiles): wave = pixel quarter: fragments 2 wave, 2 wave + 1, every subtile (<= 3).
template <int PT>
struct ChainLane {
  int prow[PT], pcol[PT], pimg[PT];
  bool pval[PT];
  unsigned act_lane[PT], px_piece[PT];
  int l31, hi;
};

// tuning aid (DV_CHAIN_PROF): shader-clock sums per computing wave
struct ChainProf {
  unsigned long long wait_b = 0, mfma = 0, wait_e = 0, epi_lds = 0, epi_hbm = 0, setup = 0;
};
__device__ __forceinline__ unsigned long long chain_clock() { return __builtin_amdgcn_s_memtime(); }

// One layer on one tile for one computing wave: NB cout subtiles x PT pixel fragments (minus
// SKIP), NT taps per chunk.  The accumulators are local to this instantiation (a switch over tap
// counts around a shared accumulator array made the register allocator spill).
template <int NB, int PT, int NT, int KW, int SKIP, bool PROF>
__device__ __forceinline__ unsigned chain_layer(const ChainArgs& p, const ChainLayer& L, bool last, char* smem,
                                                const ChainLane<PT>& c, int sub_base, int t_lo,
                                                const unsigned (&m_in)[PT], int n0, unsigned step, ChainProf& prof) {
  // everything the chunk loop needs sits in registers before it starts: the barriers are asm
  // statements with a memory clobber, anything still in memory would be re-read after each
  unsigned m[PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) m[pt] = m_in[pt];
  const int Gw = p.G * p.w;
  const int n_chunks = L.n_chunks;
  const unsigned ring0 = p.act_bytes, slot_bytes = p.slot_bytes;
  const unsigned zero_addr = p.act_bytes + 2 * p.slot_bytes;
  const unsigned chunk_lds = static_cast<unsigned>(2 * p.tpx * 16);      // one 16-channel chunk of the tile
  const unsigned group_lds = static_cast<unsigned>(p.tpx * 16);
  const unsigned a_tap_stride = static_cast<unsigned>(2 * L.cout_pad * 16);
  // first tap of the walk relative to the output pixel, and the stride between taps (chain_step)
  unsigned b_tap_stride, first_off;
  if (KW == 0) {
    const int pad = (L.kh * L.kw - 1) >> 1;
    b_tap_stride = static_cast<unsigned>(L.kw > 1 ? 16 : Gw * 16);
    first_off = static_cast<unsigned>(t_lo - pad) * b_tap_stride;
  } else {
    b_tap_stride = static_cast<unsigned>(Gw * 16);
    first_off = static_cast<unsigned>(-(((L.kh - 1) >> 1) * Gw + ((L.kw - 1) >> 1)) * 16);
  }
  unsigned b0[PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) b0[pt] = c.act_lane[pt] + first_off;
  const unsigned a_lane = static_cast<unsigned>((c.hi * L.cout_pad + sub_base * 32 + c.l31) * 16) +
                          static_cast<unsigned>(t_lo) * a_tap_stride;
  float16_t acc[NB][PT];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nb][pt][i] = 0.f;

  unsigned long long t0 = 0;
  if (PROF) t0 = chain_clock();
  // one chunk; PRE: tap 0's pixel fragments travel while the wave waits at the barrier (not for a
  // layer's first chunk: its barrier is what makes the previous layer's output visible).  The
  // first chunk is peeled off the loop -- a branch on `cc` around two instantiations inside the
  // loop made the register allocator spill the accumulators.
  auto chunk = [&](int cc, auto pre_tag) {
    constexpr bool PRE = decltype(pre_tag)::value;
    unsigned b_addr[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) b_addr[pt] = b0[pt] + static_cast<unsigned>(cc) * chunk_lds;
    half8_t b_pre[PT];
    if (PRE) {
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        b_pre[pt] = *reinterpret_cast<const half8_t*>(smem + (m[pt] & 1u ? b_addr[pt] : zero_addr));
      }
    }
    barrier_after_lds();   // B(l, cc): this chunk's weight slab (and, first chunk, the tile) landed
    if (PROF) {
      const unsigned long long t = chain_clock();
      prof.wait_b += t - t0;
      t0 = t;
    }
    const unsigned a_addr = ring0 + (step & 1u) * slot_bytes + a_lane;
    chain_step<NB, PT, NT, KW, SKIP, PRE>(smem, a_addr, a_tap_stride, b_addr, b_tap_stride, m, zero_addr, b_pre, acc);
    if (PROF) {
      asm volatile("" : "+v"(acc[NB - 1][0]));   // the step's MFMAs are issued before the clock is read
      const unsigned long long t = chain_clock();
      prof.mfma += t - t0;
      t0 = t;
    }
    ++step;
  };
  chunk(0, std::false_type{});
  for (int cc = 1; cc < n_chunks; ++cc) chunk(cc, std::true_type{});
  // the folded BatchNorm shifts of this wave's couts: requested now, in flight across the barrier
  float4_t sh[NB][4];
  {
    const float* sp = L.shift + (sub_base * 32 + 4 * c.hi);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int q = 0; q < 4; ++q) sh[nb][q] = *reinterpret_cast<const float4_t*>(sp + nb * 32 + 8 * q);
  }
  barrier_after_lds();     // E(l): every wave is done reading this layer's input
  if (PROF) {
    const unsigned long long t = chain_clock();
    prof.wait_e += t - t0;
    t0 = t;
  }
  const int cout = L.cout;
  if (!last) {
    // the layer's output replaces its input in LDS: [group][pixel][8]
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int cbase = (sub_base + nb) * 32;
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        if (!chain_tile<SKIP>(nb, pt)) continue;
        uint4_t piece[2];
        chain_pieces(acc[nb][pt], sh[nb], piece);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int group = cbase / 8 + 2 * t + c.hi;
          if (group * 8 < cout) {
            *reinterpret_cast<uint4_t*>(smem + static_cast<unsigned>(group) * group_lds + c.px_piece[pt]) = piece[t];
          }
        }
      }
    }
  } else {
    // the last layer goes to HBM, straight into the block's concat buffer
    const unsigned gstride = static_cast<unsigned>(p.og.hp * p.og.wp);
    uint4_t* outp = reinterpret_cast<uint4_t*>(p.out);
    unsigned obase[PT];
    bool ok[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const int n = n0 + c.pimg[pt];
      ok[pt] = c.pval[pt] && n < p.N;
      obase[pt] = static_cast<unsigned>(((n * p.og.groups + p.out_goff) * p.og.hp + c.prow[pt] + p.og.halo) * p.og.wp +
                                        c.pcol[pt] + p.og.halo);
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int cbase = (sub_base + nb) * 32;
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        if (!chain_tile<SKIP>(nb, pt)) continue;
        uint4_t piece[2];
        chain_pieces(acc[nb][pt], sh[nb], piece);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int group = cbase / 8 + 2 * t + c.hi;
          if (ok[pt] && group * 8 < cout) outp[obase[pt] + static_cast<unsigned>(group) * gstride] = piece[t];
        }
      }
    }
  }
  if (PROF) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t = chain_clock();
    (last ? prof.epi_hbm : prof.epi_lds) += t - t0;
  }
  return step;
}

// the layer's barriers without any matrix work (a wave whose pixels are all padding)
__device__ __forceinline__ unsigned chain_layer_idle(const ChainLayer& L, unsigned step) {
  const int n_chunks = L.n_chunks;
  for (int cc = 0; cc < n_chunks; ++cc, ++step) barrier_after_lds();
  barrier_after_lds();
  return step;
}

// Positions t of a `taps`-long filter axis that meet the map for a pixel at `pos` of `lim`:
// t + pos - pad in [0, lim).
__device__ __forceinline__ unsigned chain_axis_mask(int pos, int lim, int taps, int pad) {
  const int lo = max(0, pad - pos), hi = min(taps - 1, lim - 1 + pad - pos);
  return hi >= lo ? ((2u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
}

// Taps (row-major kh x kw) of the filter that meet the map for the pixel at (row, col).
__device__ __forceinline__ unsigned chain_tap_mask(const ChainArgs& p, const ChainLayer& L, int row, int col, bool valid) {
  if (!valid) return 0u;
  const unsigned rows = chain_axis_mask(row, p.h, L.kh, (L.kh - 1) >> 1);
  const unsigned cols = chain_axis_mask(col, p.w, L.kw, (L.kw - 1) >> 1);
  unsigned m = 0;
  for (int ty = 0; ty < L.kh; ++ty) {
    if ((rows >> ty) & 1u) m |= cols << (ty * L.kw);
  }
  return m;
}

template <int PT, bool PROF>
__device__ __forceinline__ void chain_compute(const ChainArgs& p, char* smem, int wave, int lane) {
  ChainProf prof;
  unsigned long long t_setup = 0;
  // PT 3: two pixel halves x two cout halves; PT 2: four pixel quarters
  const int ph = PT == 3 ? wave & 1 : wave, ch = PT == 3 ? wave >> 1 : 0;
  ChainLane<PT> c;
  c.l31 = lane & 31;
  c.hi = lane >> 5;
  const int Gw = p.G * p.w, T = Gw * p.h;
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int px = (ph * PT + pt) * 32 + c.l31;
    c.pval[pt] = px < T;
    const int q = c.pval[pt] ? px : 0;
    c.prow[pt] = q / Gw;
    const int rem = q - c.prow[pt] * Gw;
    c.pimg[pt] = rem / p.w;
    c.pcol[pt] = rem - c.pimg[pt] * p.w;
    c.px_piece[pt] = static_cast<unsigned>(px) * 16u;
    c.act_lane[pt] = static_cast<unsigned>(c.hi * p.tpx + px) * 16u;
  }
  const unsigned zero_addr = p.act_bytes + 2 * p.slot_bytes;
  if (wave == 0 && lane < 4) *reinterpret_cast<unsigned*>(smem + zero_addr + lane * 4) = 0u;

  // Per layer, once per kernel: the taps the wave walks (one-dimensional filters: the union of
  // its lanes' taps, from wave-wide ballots; two-dimensional ones: all) and its share of the cout
  // subtiles, packed t_lo | nt << 4 | nbw << 10 | skip << 14 | sub_base << 16.
  unsigned cfg[kChainMaxLayers];
#pragma unroll
  for (int l = 0; l < kChainMaxLayers; ++l) {
    cfg[l] = 0;
    if (l < p.n_layers) {
      const ChainLayer& L = p.L[l];
      const int taps = L.kh * L.kw;
      int t_lo = 0, nt = taps;
      if (L.kh == 1 || L.kw == 1) {
        unsigned any = 0;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) any |= chain_tap_mask(p, L, c.prow[pt], c.pcol[pt], c.pval[pt]);
        unsigned wave_any = 0;
#pragma unroll
        for (int t = 0; t < kChainMaxTaps; ++t) {
          if (__builtin_amdgcn_ballot_w64((any >> t) & 1u) != 0ull) wave_any |= 1u << t;
        }
        t_lo = wave_any ? __builtin_ctz(wave_any) : 0;
        nt = wave_any ? 32 - __builtin_clz(wave_any) - t_lo : 0;
        // compiled tap counts: 3, 5, 6, 7 -- a range in between is widened (the extra taps read
        // zeros through the lane masks, which is exact)
        while (nt != 0 && nt != 3 && nt != 5 && nt != 6 && nt != 7) {
          if (t_lo + nt < taps) {
            ++nt;
          } else {
            --t_lo;
            ++nt;
          }
        }
      } else {
        bool any_px = false;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) any_px = any_px || c.pval[pt];
        if (__builtin_amdgcn_ballot_w64(any_px) == 0ull) nt = 0;
      }
      const int subs = L.cout_pad >> 5;
      int sub_base = 0, nbw = subs, skip = 0;
      if (PT == 3) {
        if (subs == 5) {          // 8 + 7 tiles (chain_tile)
          nbw = 3;
          sub_base = ch ? 2 : 0;
          skip = ch ? 2 : 1;
        } else {
          sub_base = ch ? (subs + 1) >> 1 : 0;
          nbw = ch ? subs >> 1 : (subs + 1) >> 1;
        }
      }
      cfg[l] = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(t_lo) | static_cast<unsigned>(nt) << 4 |
                                              static_cast<unsigned>(nbw) << 10 | static_cast<unsigned>(skip) << 14 |
                                              static_cast<unsigned>(sub_base) << 16);
    }
  }

  unsigned step = 0;
  for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
    const int n0 = max(0, min(tile * p.G, p.N - p.G));
    for (int l = 0; l < p.n_layers; ++l) {
      if (PROF) t_setup = chain_clock();
      const ChainLayer& L = p.L[l];
      const bool last = l + 1 == p.n_layers;
      const unsigned cf = l == 0 ? cfg[0] : l == 1 ? cfg[1] : l == 2 ? cfg[2] : cfg[3];
      const int t_lo = cf & 15, nt = (cf >> 4) & 63, nbw = (cf >> 10) & 15, skip = (cf >> 14) & 3;
      const int sub_base = cf >> 16;
      unsigned m[PT];
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) m[pt] = chain_tap_mask(p, L, c.prow[pt], c.pcol[pt], c.pval[pt]) >> t_lo;
      if (PROF) prof.setup += chain_clock() - t_setup;
#define DV_CHAIN_CASE(NB_, NT_, KW_, SKIP_) \
  case SKIP_ * 1024 + NB_ * 64 + NT_: \
    step = chain_layer<NB_, PT, NT_, KW_, SKIP_, PROF>(p, L, last, smem, c, sub_base, t_lo, m, n0, step, prof); \
    break;
      if (PT == 3) {
        switch (skip * 1024 + nbw * 64 + nt) {   // wave-uniform; one-dimensional filters
          DV_CHAIN_CASE(2, 3, 0, 0) DV_CHAIN_CASE(2, 5, 0, 0) DV_CHAIN_CASE(2, 6, 0, 0) DV_CHAIN_CASE(2, 7, 0, 0)
          DV_CHAIN_CASE(3, 3, 0, 0) DV_CHAIN_CASE(3, 5, 0, 0) DV_CHAIN_CASE(3, 6, 0, 0) DV_CHAIN_CASE(3, 7, 0, 0)
          DV_CHAIN_CASE(3, 3, 0, 1) DV_CHAIN_CASE(3, 5, 0, 1) DV_CHAIN_CASE(3, 6, 0, 1) DV_CHAIN_CASE(3, 7, 0, 1)
          DV_CHAIN_CASE(3, 3, 0, 2) DV_CHAIN_CASE(3, 5, 0, 2) DV_CHAIN_CASE(3, 6, 0, 2) DV_CHAIN_CASE(3, 7, 0, 2)
          default: step = chain_layer_idle(L, step); break;   // nt == 0 (the host admits only these shapes)
        }
      } else {
        switch (nbw * 64 + nt) {                  // 3x3 and 5x5 filters
          DV_CHAIN_CASE(2, 9, 3, 0) DV_CHAIN_CASE(3, 9, 3, 0) DV_CHAIN_CASE(2, 25, 5, 0) DV_CHAIN_CASE(3, 25, 5, 0)
          default: step = chain_layer_idle(L, step); break;
        }
      }
#undef DV_CHAIN_CASE
    }
  }
  if (PROF && lane == 0 && p.prof != nullptr) {
    unsigned long long* dst = p.prof + (static_cast<size_t>(blockIdx.x) * 4 + wave) * 8;
    dst[0] = prof.wait_b;
    dst[1] = prof.mfma;
    dst[2] = prof.wait_e;
    dst[3] = prof.epi_lds;
    dst[4] = prof.epi_hbm;
    dst[5] = prof.setup;
  }
}

// ------------------------------------------------------------------ moving waves (4-7)
// The weight slab of the next chunk and the next tile's input by LDS-DMA, one chunk of lead.  (A
// register-staged variant with two chunks of lead -- global -> VGPR -> ds_write -- measured 1-2 %
// SLOWER: the phase profile shows the computing waves, not the loaders, set the pace.)
__device__ __forceinline__ void chain_move(const ChainArgs& p, char* smem, int lw, int lane) {
  const int Gw = p.G * p.w, T = Gw * p.h;
  const int parts = p.tpx >> 6;                 // 64-pixel parts of a channel group of the tile
  // source offset of tile pixel part*64 + lane, relative to (first image of the tile, group 0)
  unsigned src[4];
#pragma unroll
  for (int part = 0; part < 4; ++part) {
    const int px = part * 64 + lane;
    const int q = px < T ? px : 0;
    const int row = q / Gw, rem = q - row * Gw;
    const int img = rem / p.w, col = rem - img * p.w;
    src[part] = static_cast<unsigned>(img) * p.in_img_bytes +
                static_cast<unsigned>(((row + p.ig.halo) * p.ig.wp + col + p.ig.halo) * 16);
  }
  const unsigned plane_bytes = static_cast<unsigned>(p.ig.hp * p.ig.wp * 16);
  const unsigned ring0 = p.act_bytes;
  const int in_groups = p.L[0].n_chunks * 2;

  auto issue_weights = [&](int l, int cc, unsigned slot) {
    const ChainLayer& L = p.L[l];
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(uniform_ptr(reinterpret_cast<const char*>(L.w) + static_cast<size_t>(cc) * L.slab_bytes)),
        0, L.slab_bytes, 0x00020000);
    const int pieces = static_cast<int>(L.slab_bytes >> 10);
    for (int j = lw; j < pieces; j += 4) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(smem + ring0 + slot * p.slot_bytes + j * 1024), 16,
                                               lane * 16, j * 1024, 0, 0);
    }
  };
  auto issue_tile = [&](int tile) {
    const int n0 = max(0, min(tile * p.G, p.N - p.G));
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(uniform_ptr(reinterpret_cast<const char*>(p.in) + static_cast<size_t>(n0) * p.in_img_bytes)),
        0, 0x7fffffff, 0x00020000);
    for (int g = 0; g < in_groups; ++g) {
#pragma unroll
      for (int part = 0; part < 4; ++part) {
        if (part < parts && ((g * parts + part) & 3) == lw) {   // wave-uniform
          __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(smem + (g * p.tpx + part * 64) * 16), 16, src[part],
                                                   g * plane_bytes, 0, 0);
        }
      }
    }
  };

  int tile = blockIdx.x;
  if (tile >= p.n_tiles) return;
  issue_weights(0, 0, 0u);
  issue_tile(tile);
  unsigned step = 0;
  for (; tile < p.n_tiles; tile += gridDim.x) {
    const bool more_tiles = tile + static_cast<int>(gridDim.x) < p.n_tiles;
    for (int l = 0; l < p.n_layers; ++l) {
      const int n_chunks = p.L[l].n_chunks;
      for (int cc = 0; cc < n_chunks; ++cc, ++step) {
        barrier_after_dma();   // B(l, cc): everything issued so far has landed
        // the slot the computing waves left at this barrier takes the next chunk's slab
        int nl = l, ncc = cc + 1;
        if (ncc == n_chunks) {
          ncc = 0;
          nl = l + 1;
        }
        bool has_next = true;
        if (nl == p.n_layers) {
          nl = 0;
          has_next = more_tiles;
        }
        if (has_next) issue_weights(nl, ncc, (step + 1u) & 1u);
      }
      barrier_only();          // E(l): the DMAs in flight stay in flight
      if (l == p.n_layers - 1 && more_tiles) issue_tile(tile + gridDim.x);
    }
  }
}

template <int PT, bool PROF>
__global__ __launch_bounds__(CH_THREADS, 1) void chain_kernel(ChainArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  if (static_cast<int>(blockIdx.x) >= p.n_tiles) return;
  if (wave < 4) {
    chain_compute<PT, PROF>(p, smem, wave, lane);
  } else {
    chain_move(p, smem, wave - 4, lane);
  }
}

}  // namespace

size_t chain_lds_bytes(const ChainArgs& a) {
  return static_cast<size_t>(a.act_bytes) + 2 * static_cast<size_t>(a.slot_bytes) + 16;
}

template <int PT, bool PROF>
static void launch_chain_as(const ChainArgs& a, int grid, hipStream_t stream) {
  static const bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<PT, PROF>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)attr;
  hipLaunchKernelGGL((chain_kernel<PT, PROF>), dim3(grid), dim3(CH_THREADS), chain_lds_bytes(a), stream, a);
}

void launch_chain(const ChainArgs& a, int blocks, hipStream_t stream) {
  int grid = a.n_tiles < blocks ? a.n_tiles : blocks;
  if (grid < 1) grid = 1;
  const bool prof = a.prof != nullptr;
  if (a.tpx == 192) {
    if (prof) launch_chain_as<3, true>(a, grid, stream); else launch_chain_as<3, false>(a, grid, stream);
  } else {
    if (prof) launch_chain_as<2, true>(a, grid, stream); else launch_chain_as<2, false>(a, grid, stream);
  }
}

}  // namespace dv
