// model.hip -- Inception-v3 call_variants classifier on gfx950 (MI355X).
//
// Replaces the model forward of deepvariant/call_variants.py:904-932
// (keras_modeling.inceptionv3, deepvariant/keras_modeling.py:246-336; graph =
// tf_keras InceptionV3(include_top=False, pooling='avg')).
//
// Data layout in HBM: activations are fp16 in the channel-blocked, zero-haloed layout
// [N][C/8][H+2h][W+2h][8] ("C8"): the 8 channels an MFMA fragment needs for one pixel
// are one 16-byte piece, and consecutive pixels of a row are consecutive
// pieces, so every fragment load and every store of a 32-pixel tile is one
// contiguous 512-byte run.  One buffer per graph tensor; concat outputs are
// written in place (each branch's last conv stores at its channel-group offset
// of the block's output buffer -- no concat kernel).  BatchNorm
// (scale=False, eps=1e-3, moving statistics) is folded into the fp16 conv
// weights and an fp32 per-channel shift at load time.
//
// Kernels (HISTORY.md 4.2 has the measurements behind each choice)
//   conv_mfma_kernel<NB,PT>   implicit-GEMM conv + shift + ReLU on v_mfma_f32_32x32x16_f16:
//                             D[cout][pixel] = sum_k W[cout][k] X[k][pixel]; a wave owns
//                             PT*32 pixels x NB*32 couts; weights stream through LDS in
//                             slabs of 8 K-chunks, pixel fragments go global -> VGPR;
//                             sibling 1x1 heads share one launch over their concatenated
//                             couts
//   conv_first_u8_kernel<PT>  first 3x3/2 conv straight from the uint8 pileup tensor
//                             ((x-128)/128 in registers)
//   conv_pool1x1_kernel<NB>   1x1 conv whose input is max-pooled (3x3/2) on the fly
//   preprocess_kernel         uint8 HWC -> fp16 C8, only for inputs with > 8 channels
//   maxpool3s2_kernel / avgpool3s1_kernel   (avg excludes padding, optional shift + ReLU)
//   head_kernel               global average pool + Dense(3) + softmax in fp32
#include <algorithm>
#include <map>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <type_traits>

#include <hip/hip_fp16.h>

#include "dv_internal.h"
#include "conv_common.h"
#include "imgconv.h"
#include "chain.h"
#include "stem_fused.h"
#include "calib.h"

using namespace dv::convk;

namespace {

// Examples per stem pass.  Measured on MI355X (round 1): sub-batching the stem to
// keep its hand-offs in the 256 MB Infinity Cache (64/128/256) LOSES 5-19 % at
// 2 K examples per launch -- the stem is not HBM-bound and the extra launches
// cost ~5 us each -- so it is off by default; DV_STEM_SB enables it for tuning.
static int stem_sub_batch() {
  static const int v = getenv("DV_STEM_SB") ? std::max(1, atoi(getenv("DV_STEM_SB"))) : (1 << 30);
  return v;
}
constexpr int kFirstUnroll = 5;  // conv_first_u8_kernel: chunks of a 3x3 filter (2 taps per chunk)
constexpr int kSlabChunks = 8;   // K chunks (of 16 channels) per weight slab
// Pixel-operand prefetch depth, in chunks, per tile shape.  Measured on MI355X: 8 or 16
// instead of 4 changes nothing (+-1 %) for thin, mid or big tiles -- the queue is not what
// the waves wait for (HISTORY.md 7) -- so every shape uses 4.
constexpr int prefetch_depth(int /*nb*/, int /*pt*/) { return 4; }

// Wave-uniform walk over the K chunks kept in SGPRs and advanced with selects
// only -- no memory, no branches.  K order is channel-chunk major, filter tap
// minor: the KH*KW taps of one 16-channel chunk are consecutive, so a block
// re-reads the same two channel-group planes (a few KB incl. halo) KH*KW times
// back to back and the vector L1 serves all but the first pass.
// off = byte offset of (group 2*cc, kh, kw) relative to (group 0, ih0, iw0).
struct ChunkWalk {
  int kh, kw;
  unsigned tap_off;    // (kh*wp + kw) * 16
  unsigned chunk_off;  // cc * chunk_stride
  __device__ __forceinline__ unsigned off() const { return chunk_off + tap_off; }
  __device__ __forceinline__ void advance(const ConvArgs& p) {
    const bool row_end = ++kw == p.KW;
    kw = row_end ? 0 : kw;
    tap_off += row_end ? static_cast<unsigned>((p.ig.wp - p.KW + 1) * 16) : 16u;
    kh += row_end ? 1 : 0;
    const bool taps_end = kh == p.KH;
    kh = taps_end ? 0 : kh;
    tap_off = taps_end ? 0u : tap_off;
    chunk_off += taps_end ? p.chunk_stride : 0u;
  }
};

// One weight slab of R (<= 8) K-chunks: straight-line code, no branches, so the
// compiler's s_waitcnt insertion keeps the kPrefetch-deep load pipeline intact.
// Pixel fragments: voffset = per-lane base (VGPR, fixed for the whole kernel),
// soffset = chunk offset (SGPR): ZERO vector ALU work per load.  Chunks past the
// end of K simply read the next bytes of the (larger) input tensor or hit the
// descriptor's range check; their values are never used.
// SPLIT (ConvArgs::split): weight chunks come in (hi, lo) pairs -- W = W_hi + W_lo, both fp16 --
// that multiply the SAME pixel fragment: chunk j of the slab uses pixel slot (S0 + j) / 2 and
// the slot is refilled after the lo half.  The products are exact and the accumulator is fp32,
// so the layer sees 22-bit weights for one extra MFMA and one extra ds_read per fragment.
// State of the side max-pool (ConvArgs::side_pool_out) of one wave: the running maximum of the
// current 16-channel chunk's window pieces and where the finished ones go.
template <int PT>
struct SidePool {
  half8_t best[PT];
  int tap, cc;          // position in the K walk (wave-uniform): tap of the chunk, channel chunk
  int cc_mod;           // cc % n_tiles: the cout tiles of a pixel block load the same fragments and share
  int n_tiles, my_tile; // the pool's chunks round robin (tile t stores the chunks with cc % n_tiles == t)
  int taps, n_cc;
  unsigned at[PT];      // piece index of (n, group side_pool_goff + hi, oh, ow) in the pooled tensor
  bool ok[PT];
  unsigned gstride2;    // two channel groups (one chunk) further
  uint4_t* out;
};

template <int NB, int PT, int R, int S0 = 0, bool SPLIT = false, bool POOL = false>
__device__ __forceinline__ void conv_slab(const ConvArgs& p, const __amdgpu_buffer_rsrc_t rsrc,
                                          const _Float16* wslab, ChunkWalk& walk,
                                          const unsigned (&base)[PT],
                                          uint4_t (&xf)[prefetch_depth(NB, PT)][PT],
                                          float16_t (&acc)[NB][PT], SidePool<PT>* sp = nullptr) {
  constexpr int BN = NB * 32;
  constexpr int kPrefetch = prefetch_depth(NB, PT);
  // Weight fragments are double buffered in registers: the ds_reads of chunk
  // j+1 are issued before the MFMAs of chunk j, whose 32*NB*PT cycles cover the
  // LDS latency.
  half8_t wf[2][NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    wf[0][nb] = *reinterpret_cast<const half8_t*>(wslab + (nb * 32) * 8);
  }
#pragma unroll
  for (int j = 0; j < R; ++j) {
    // DV_ABLATE_* (tools/ablate_conv.sh): timing ablations, results are WRONG by construction --
    // _W reuses the first chunk's weight fragments, _X never refills the pixel fragments,
    // _LOOP skips the K loop, _EPI (conv_common.h) suppresses the output stores.
#ifndef DV_ABLATE_W
    if (j + 1 < R) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        wf[(j + 1) & 1][nb] = *reinterpret_cast<const half8_t*>(
            wslab + (j + 1) * BN * kChunk + (nb * 32) * 8);
      }
    }
#else
    if (j + 1 < R) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) wf[(j + 1) & 1][nb] = wf[j & 1][nb];
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
    constexpr int kDiv = SPLIT ? 2 : 1;
    const int slot = ((S0 + j) / kDiv) % kPrefetch;
    half8_t xh[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) xh[pt] = __builtin_bit_cast(half8_t, xf[slot][pt]);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        acc[nb][pt] =
            __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j & 1][nb], xh[pt], acc[nb][pt], 0, 0, 0);
      }
    }
    if constexpr (POOL) {
      // this step's fragments ARE window pieces of the sibling max-pool (K order: channel chunk
      // major, tap minor); exact, so the result equals the separate max-pool kernel's bit for bit.
      // The cout tiles of a pixel block see the same fragments: tile t keeps the chunks with
      // cc % n_tiles == t (wave-uniform), the running maximum restarts from the lowest fp16 number.
      if (sp->cc_mod == sp->my_tile) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) sp->best[pt] = __builtin_elementwise_max(sp->best[pt], xh[pt]);
      }
      if (sp->tap == sp->taps - 1) {   // wave-uniform
        if (sp->cc < sp->n_cc && sp->cc_mod == sp->my_tile) {
          const _Float16 lowest = static_cast<_Float16>(-65504.f);
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) {
            if (sp->ok[pt]) sp->out[sp->at[pt]] = __builtin_bit_cast(uint4_t, sp->best[pt]);
            sp->best[pt] = half8_t{lowest, lowest, lowest, lowest, lowest, lowest, lowest, lowest};
          }
        }
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) sp->at[pt] += sp->gstride2;
        sp->tap = 0;
        ++sp->cc;
        sp->cc_mod = sp->cc_mod + 1 == sp->n_tiles ? 0 : sp->cc_mod + 1;
      } else {
        ++sp->tap;
      }
    }
    // refill the slot just consumed with chunk (current + kPrefetch)
    if (!SPLIT || ((S0 + j) & 1)) {
#ifndef DV_ABLATE_X
      const unsigned soff = walk.off();
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        xf[slot][pt] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base[pt], soff, 0);
      }
#endif
      walk.advance(p);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// conv_slab for a WIDE input (ConvArgs::wide_in): R K-chunks, each multiplied against the hi and then the lo pixel
// fragment with the same weight fragments.  The prefetch ring of four fragments holds two chunks' (hi, lo) pairs; the
// fragment consumed is replaced by the same part of the chunk two further on.  R is even (slabs of 8 or 4 chunks), so the
// ring position is 0 at every call.
template <int NB, int PT, int R>
__device__ __forceinline__ void conv_slab_wide(const ConvArgs& p, const __amdgpu_buffer_rsrc_t rsrc,
                                               const _Float16* wslab, ChunkWalk& walk, const unsigned (&base)[PT],
                                               uint4_t (&xf)[prefetch_depth(NB, PT)][PT], float16_t (&acc)[NB][PT]) {
  constexpr int BN = NB * 32;
  static_assert(prefetch_depth(NB, PT) == 4 && R % 2 == 0, "two (hi, lo) pairs in flight");
  half8_t wf[2][NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) wf[0][nb] = *reinterpret_cast<const half8_t*>(wslab + (nb * 32) * 8);
#pragma unroll
  for (int j = 0; j < R; ++j) {
    if (j + 1 < R) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        wf[(j + 1) & 1][nb] = *reinterpret_cast<const half8_t*>(wslab + (j + 1) * BN * kChunk + (nb * 32) * 8);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int part = 0; part < 2; ++part) {
      const int slot = (2 * j + part) & 3;
      half8_t xh[PT];
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) xh[pt] = __builtin_bit_cast(half8_t, xf[slot][pt]);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          acc[nb][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j & 1][nb], xh[pt], acc[nb][pt], 0, 0, 0);
        }
      }
      const unsigned soff = walk.off() + (part ? p.lo_off : 0u);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) xf[slot][pt] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base[pt], soff, 0);
      if (part) walk.advance(p);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Blank-row skipping (ConvArgs::blank_row): the wave's PT*32 pixels x NB*32 couts copied from the
// all-blank image's response instead of computed.  Lanes l / l+32 take alternate 8-cout
// groups; all loads are issued before the first store.
template <int NB, int PT>
__device__ __forceinline__ void copy_blank_wave(const ConvArgs& p, int n_tile, const int (&pn)[PT],
                                                const int (&poh)[PT], const int (&pow_)[PT],
                                                const bool (&mvalid)[PT], int lane) {
  const ConvBranch& b = p.br[0];
  const unsigned gstride = static_cast<unsigned>(b.og.hp * b.og.wp);
  const int hi = lane >> 5;
  const uint4_t* src = reinterpret_cast<const uint4_t*>(p.blank_src);
  uint4_t* dst = reinterpret_cast<uint4_t*>(b.out);
  uint4_t v[PT][NB * 2];
  unsigned at[PT][NB * 2];
  bool ok[PT][NB * 2];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const unsigned rel = static_cast<unsigned>((poh[pt] + b.og.halo) * b.og.wp + pow_[pt] + b.og.halo);
#pragma unroll
    for (int j = 0; j < NB * 2; ++j) {
      const int g = n_tile * NB * 4 + 2 * j + hi;
      ok[pt][j] = mvalid[pt] && g * 8 < b.Cout;
      at[pt][j] = static_cast<unsigned>(b.out_goff + g) * gstride + rel;
      v[pt][j] = ok[pt][j] ? src[at[pt][j]] : uint4_t{0u, 0u, 0u, 0u};
    }
  }
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const unsigned img = static_cast<unsigned>(pn[pt] * b.og.groups) * gstride;
#pragma unroll
    for (int j = 0; j < NB * 2; ++j) {
      if (ok[pt][j]) dst[img + at[pt][j]] = v[pt][j];
    }
  }
}

// Implicit-GEMM convolution, D[cout][pixel] = sum_k W[cout][k] * X[k][pixel].
//
//  * The block's weight tile (NB*32 couts) streams through LDS in slabs of 8
//    K-chunks (128 K values), double buffered: ONE barrier per 8*NB*PT MFMAs.
//  * The pixel operand never touches LDS: an MFMA B fragment is 8 consecutive
//    channels of one pixel = one 16-byte piece of the C8 layout, loaded
//    kPrefetch chunks ahead straight into VGPRs; 32 consecutive pixels are one
//    contiguous 512-byte run.
//  * Each wave owns PT*32 pixels x all NB*32 couts of the tile: NB*PT
//    independent 32x32 accumulators keep the matrix pipe busy back to back.
//  * Epilogue: shift + ReLU, lanes l / l+32 pair their halves into 16-byte
//    pieces, stored as contiguous 512-byte runs (no LDS).
template <int NB, int PT, int MINB = (NB * PT >= 8 ? 1 : 2), int SLAB = kSlabChunks, int WAVES = 4,
          bool SPLIT = false, bool SIDE_POOL = false, bool AVG = false, bool WIDE = false>
__global__ __launch_bounds__(WAVES * 64, MINB) void conv_mfma_kernel(ConvArgs p) {
  constexpr int BN = NB * 32;
  constexpr int kThreads = WAVES * 64;   // (WAVES = 8: tuning experiment DV_CONV_W8, HISTORY.md 7)
  constexpr int SLAB_HALFS = SLAB * BN * kChunk;
  constexpr int SLAB_PIECES = SLAB_HALFS / 8;            // 16-byte pieces
  constexpr int W_PER_THREAD = SLAB_PIECES / kThreads;   // = 2 * NB with four waves
  static_assert(SLAB_PIECES % kThreads == 0, "slab does not split evenly over the block");
  constexpr int kPrefetch = prefetch_depth(NB, PT);
  extern __shared__ __attribute__((aligned(16))) _Float16 smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  // 1-D grid.  Blocks are dispatched round-robin over the 8 XCDs (block b ->
  // XCD b % 8); remap so that logically consecutive blocks -- the cout tiles of
  // the same pixel tile, which re-read the same input -- share an XCD's L2.
  const int nwg = gridDim.x;
  const int xq = nwg >> 3, xr = nwg & 7;
  const int xcd = blockIdx.x & 7;
  int xi = blockIdx.x >> 3;
  if (p.cu_pair) {
    // An XCD hands its blocks to its 32 CUs in turn: block xi runs on CU xi % 32, and with two
    // blocks per CU the ones 32 apart share a CU.  Give those two consecutive logical numbers.
    const int cnt = xcd < xr ? xq + 1 : xq;
    if (xi < cnt / 64 * 64) {
      const int w = xi & 63;
      xi = (xi & ~63) + (w & 31) * 2 + (w >> 5);
    }
  }
  const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
  const int n_tile = logical % p.n_tiles;
  int pix_block = logical / p.n_tiles;
  int band_r = 0;  // row-band mode: the output row of every pixel of this block
  if (p.band) {
    band_r = pix_block % p.band;   // rows minor: the H blocks reading the same images are neighbours
    pix_block /= p.band;
  }
  const int m_block = pix_block * (32 * PT * WAVES);

  // Buffer offsets are 32 bit, tensors are not (8 K examples x 1.4 MB): every wave
  // addresses the input relative to the first example it touches (n0), through its
  // own descriptor -- a wave's 64 pixels never span more than a few hundred KB.
  int n0 = 0;
  unsigned base[PT];   // byte offset of (n - n0, group lane>>5, ih0, iw0) in the input
  int pn[PT], poh[PT], pow_[PT];  // output coordinates, turned into addresses per branch
  bool mvalid[PT];
  const int ohow = p.OH * p.OW;
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int m = m_block + (wave * PT + pt) * 32 + (lane & 31);
    int n, pix, oh, ow, iy;
    if constexpr (AVG) {   // image-aligned tiles (ConvArgs::tile_g): slot -> (map of the block, pixel of the map)
      int il;
      divmod_small((wave * PT + pt) * 32 + (lane & 31), p.tile_p, p.rcp_tile_p, il, pix);
      n = pix_block * p.tile_g + il;
      mvalid[pt] = il < p.tile_g && n < p.N;
      n = min(n, p.N - 1);
      divmod_small(pix, p.OW, p.rcp_ow, oh, ow);
      iy = oh * p.stride - p.pad_h + p.ig.halo;
    } else if (p.band) {  // m runs over (example, column) of row band_r; taps start at map row 0
      mvalid[pt] = m < p.N * p.OW;
      divmod_small(mvalid[pt] ? m : 0, p.OW, p.rcp_ow, n, ow);
      oh = band_r;
      iy = p.ig.halo;
    } else {
      mvalid[pt] = m < p.M;
      divmod_small(mvalid[pt] ? m : 0, ohow, p.rcp_ohow, n, pix);
      divmod_small(pix, p.OW, p.rcp_ow, oh, ow);
      iy = oh * p.stride - p.pad_h + p.ig.halo;
    }
    const int ix = ow * p.stride - p.pad_w + p.ig.halo;
    if (pt == 0) n0 = __builtin_amdgcn_readfirstlane(n);  // lane 0 holds the wave's first pixel
    base[pt] = mvalid[pt]
                   ? static_cast<unsigned>(((((n - n0) * p.ig.groups + (lane >> 5)) * p.ig.hp + iy) *
                                                p.ig.wp + ix) * 16)
                   : 0x80000000u;  // beyond the descriptor's range: reads as zero
    pn[pt] = n;
    poh[pt] = oh;
    pow_[pt] = ow;
  }

  // Blank-row skipping (opt-in): a pixel range that lies in ONE example, from a row at or past
  // that example's first blank-determined row, is copied instead of computed.  Decided for the
  // whole block (no slab traffic, no barriers) and per wave (the wave keeps its share of the
  // weight-slab copies and the barriers, but issues no pixel loads and no MFMAs).
  bool wave_blank = false;
  if (p.blank_row != nullptr) {
    auto range_blank = [&](int m_lo, int m_hi) -> bool {  // pixels [m_lo, m_hi), uniform arguments
      m_hi = min(m_hi, p.M);
      if (m_lo >= m_hi) return false;
      int nf, pf, nl, pl, ohf, owf;
      divmod_small(m_lo, ohow, p.rcp_ohow, nf, pf);
      divmod_small(m_hi - 1, ohow, p.rcp_ohow, nl, pl);
      divmod_small(pf, p.OW, p.rcp_ow, ohf, owf);
      return nf == nl && ohf >= p.blank_row[nf];
    };
    if (range_blank(m_block, m_block + 32 * PT * WAVES)) {   // block-uniform, before any barrier
      bool wanted = true;   // rows nobody reads are not even copied (ConvArgs::blank_need)
      if (p.blank_need != nullptr) {
        int nf, pf, ohf, owf;
        divmod_small(m_block, ohow, p.rcp_ohow, nf, pf);
        divmod_small(pf, p.OW, p.rcp_ow, ohf, owf);
        wanted = ohf < p.blank_need[nf];
      }
      if (wanted) copy_blank_wave<NB, PT>(p, n_tile, pn, poh, pow_, mvalid, lane);
      return;
    }
    const int m_wave = m_block + wave * (32 * PT);
    wave_blank = __builtin_amdgcn_readfirstlane(range_blank(m_wave, m_wave + 32 * PT) ? 1 : 0) != 0;
  }

  const size_t in_off = static_cast<size_t>(n0) * p.img_bytes;
  const size_t in_left = p.in_bytes - in_off;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.in) + in_off), 0,
      static_cast<unsigned>(in_left < 0x7fffffffu ? in_left : 0x7fffffffu), 0x00020000);

  // ---- weight slabs: global -> registers -> LDS ---------------------------
  const uint4* wsrc = reinterpret_cast<const uint4*>(p.w) +
                      static_cast<size_t>(band_r * p.n_tiles + n_tile) * p.n_slabs * (kSlabChunks * BN * 2);
  uint4_t wreg[W_PER_THREAD];
#define DV_LOAD_SLAB(s_)                                                                   \
  {                                                                                        \
    const uint4_t* src_ = reinterpret_cast<const uint4_t*>(wsrc) +                         \
                          static_cast<size_t>(s_) * SLAB_PIECES + tid;                     \
    _Pragma("unroll") for (int j_ = 0; j_ < W_PER_THREAD; ++j_) wreg[j_] =                \
        src_[j_ * kThreads];                                                           \
  }
#define DV_STORE_SLAB(buf_)                                                                \
  {                                                                                        \
    uint4_t* dst_ = reinterpret_cast<uint4_t*>(smem + (buf_) * SLAB_HALFS) + tid;          \
    _Pragma("unroll") for (int j_ = 0; j_ < W_PER_THREAD; ++j_) dst_[j_ * kThreads] = \
        wreg[j_];                                                                          \
  }

  uint4_t xf[kPrefetch][PT];
  float16_t acc[NB][PT];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nb][pt][i] = 0.f;

  ChunkWalk walk{0, 0, 0u, 0u};
  DV_LOAD_SLAB(0)
#pragma unroll
  for (int d = 0; d < kPrefetch; ++d) {  // chunks 0 .. kPrefetch-1 (wide input: the (hi, lo) pairs of chunks 0 and 1)
    const unsigned soff = walk.off() + (WIDE && (d & 1) ? p.lo_off : 0u);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      xf[d][pt] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base[pt], soff, 0);
    }
    if (!WIDE || (d & 1)) walk.advance(p);
  }
  DV_STORE_SLAB(0)
  __syncthreads();

  // LDS image of a chunk: [k-group g = lane>>5][cout][8 halfs] -> each half-wave
  // reads 512 contiguous bytes: conflict-free for ds_read_b128 (a [cout][16]
  // image is 2-way conflicted: measured SQ_LDS_BANK_CONFLICT ~ LDS active).
  const int frag_off = (lane >> 5) * (BN * 8) + (lane & 31) * 8;  // halfs
  if (wave_blank) {   // wave-uniform: same slab copies and barriers as the computing waves
    const int n_full_b = p.n_chunks / SLAB;
    for (int s = 0; s < n_full_b; ++s) {
      const int next = s + 1 < (p.n_chunks + SLAB - 1) / SLAB ? s + 1 : s;
      DV_LOAD_SLAB(next)
      DV_STORE_SLAB((s + 1) & 1)
      __syncthreads();
    }
    copy_blank_wave<NB, PT>(p, n_tile, pn, poh, pow_, mvalid, lane);
    return;
  }
  // The K loop of one cout tile over `n_chunks` weight chunks, with (S = true) or without the
  // (hi, lo) pairing of split weights.  A split launch may mix both kinds of tile: the leading
  // ConvArgs::split_tiles cout tiles carry W_hi + W_lo (2 K chunks), the others plain weights
  // (K chunks, the first half of their slot in the packed image) -- block-uniform choice.
  SidePool<PT> side;
  auto k_loop = [&](auto split_tag, auto pool_tag, const int n_chunks) {
    constexpr bool S = decltype(split_tag)::value;
    constexpr bool P = decltype(pool_tag)::value;
    SidePool<PT>* const sp = P ? &side : nullptr;
    const int tile_slabs = (n_chunks + SLAB - 1) / SLAB;
#ifdef DV_ABLATE_LOOP
    const int n_full = n_chunks < 0 ? 1 : 0;
    const int rem = 0;
#else
    const int n_full = n_chunks / SLAB;
    const int rem = n_chunks - n_full * SLAB;
#endif
    for (int s = 0; s < n_full; ++s) {
      // The next slab's global loads are UNCONDITIONAL (the last trip re-reads its own slab
      // into the idle buffer): behind an `if` the compiler has to assume at the first
      // pixel-fragment wait that they were not issued and emits vmcnt(7) -- which, when they
      // were, drains the whole four-chunk prefetch queue at every slab start.
      const int next = s + 1 < tile_slabs ? s + 1 : s;
      DV_LOAD_SLAB(next)
      if constexpr (WIDE) {
        conv_slab_wide<NB, PT, SLAB>(p, rsrc, smem + (s & 1) * SLAB_HALFS + frag_off, walk, base, xf, acc);
      } else {
        conv_slab<NB, PT, SLAB, 0, S, P>(p, rsrc, smem + (s & 1) * SLAB_HALFS + frag_off, walk, base, xf, acc, sp);
      }
      DV_STORE_SLAB((s + 1) & 1)
      __syncthreads();
    }
    // Tail slab (n_chunks % 8 chunks), in straight-line groups of 4: K is padded
    // to a multiple of 4 chunks with zero weights (the slab image is zero there),
    // so no chunk count ever needs a branch or a register rotation inside the
    // load pipeline.  (A rolled one-chunk loop had to rotate the prefetch slots and
    // drained vmcnt(0) every chunk; per-count unrolled variants behind a switch
    // made the register allocator clone the accumulators.)
    if (rem) {
      const _Float16* wslab = smem + (n_full & 1) * SLAB_HALFS + frag_off;
      if constexpr (WIDE) {
        conv_slab_wide<NB, PT, 4>(p, rsrc, wslab, walk, base, xf, acc);
        if constexpr (SLAB > 4) {
          if (rem > 4) conv_slab_wide<NB, PT, 4>(p, rsrc, wslab + 4 * BN * kChunk, walk, base, xf, acc);
        }
      } else {
      conv_slab<NB, PT, 4, 0, S, P>(p, rsrc, wslab, walk, base, xf, acc, sp);
      if constexpr (SLAB > 4) {
        if (rem > 4) conv_slab<NB, PT, 4, 4, S, P>(p, rsrc, wslab + 4 * BN * kChunk, walk, base, xf, acc, sp);
      }
      }
    }
  };
  if constexpr (SPLIT) {
    if (n_tile < p.split_tiles) {
      k_loop(std::true_type{}, std::false_type{}, p.n_chunks);
    } else {
      k_loop(std::false_type{}, std::false_type{}, p.n_chunks >> 1);
    }
  } else if constexpr (SIDE_POOL) {   // the reduction block's 3x3 / 2 with its sibling max-pool on the side
    side.tap = 0;
    side.cc = 0;
    side.cc_mod = 0;
    side.n_tiles = p.n_tiles;
    side.my_tile = n_tile;
    side.taps = p.KH * p.KW;
    side.n_cc = p.Cin / kChunk;
    side.gstride2 = 2u * static_cast<unsigned>(p.side_pool_og.hp * p.side_pool_og.wp);
    side.out = reinterpret_cast<uint4_t*>(p.side_pool_out);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const _Float16 lowest = static_cast<_Float16>(-65504.f);
      side.best[pt] = half8_t{lowest, lowest, lowest, lowest, lowest, lowest, lowest, lowest};
      side.ok[pt] = mvalid[pt];
      side.at[pt] = static_cast<unsigned>(
          ((pn[pt] * p.side_pool_og.groups + p.side_pool_goff + (lane >> 5)) * p.side_pool_og.hp + poh[pt] +
           p.side_pool_og.halo) * p.side_pool_og.wp + pow_[pt] + p.side_pool_og.halo);
    }
    k_loop(std::false_type{}, std::true_type{}, p.n_chunks);
  } else {
    k_loop(std::false_type{}, std::false_type{}, p.n_chunks);
  }
#undef DV_LOAD_SLAB
#undef DV_STORE_SLAB

  if constexpr (AVG) {
    conv_epilogue_avg<NB, PT>(acc, p, n_tile, pix_block, pn, poh, pow_, mvalid, lane, wave, smem);
  } else {
    conv_epilogue<NB, PT>(acc, p, n_tile, pn, poh, pow_, mvalid, lane);
  }
}

// Resident-weight variant of conv_mfma_kernel for layers whose whole cout tile fits the CU's LDS
// (K * NB*32 halfs <= ~150 KB: the 3x3 80->192, short-K 1x1 heads).  HISTORY.md 7: halving a
// launch's MFMA work and pixel traffic moved it by 10 %, so what a block of the streaming
// kernel waits for is its weight slabs (global -> registers -> LDS behind a barrier per slab,
// 138 KB per 256-pixel block on the 3x3 80->192).  Here a PERSISTENT block of eight waves
// copies its cout tile's packed weights into LDS once; after that there is no barrier and no
// weight traffic at all: every wave walks its own sequence of PT*32-pixel tiles with the same
// straight-line slab code (conv_slab), the same K order and the same epilogue, so results are
// bit-identical to conv_mfma_kernel's.  Blocks b and b + 8 (same XCD, same L2) take the
// cout tiles of the same pixels.
template <int NB, int PT>
__global__ __launch_bounds__(512, 1) void conv_resident_kernel(ConvArgs p) {
  constexpr int BN = NB * 32;
  constexpr int WAVES = 8, kThreads = WAVES * 64;
  constexpr int SLAB_HALFS = kSlabChunks * BN * kChunk;
  constexpr int kPrefetch = prefetch_depth(NB, PT);
  extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int blk = blockIdx.x;
  const int n_tile = (blk >> 3) % p.n_tiles;
  const int seq = (blk & 7) + 8 * ((blk >> 3) / p.n_tiles);
  const int n_seq = 8 * ((static_cast<int>(gridDim.x) >> 3) / p.n_tiles);
  {  // the cout tile's weights: [slab][chunk][2 k-groups][cout][8], contiguous in the packed image
    const uint4_t* wsrc = reinterpret_cast<const uint4_t*>(p.w) +
                          static_cast<size_t>(n_tile) * p.n_slabs * (kSlabChunks * BN * 2);
    uint4_t* wdst = reinterpret_cast<uint4_t*>(smem);
    const int pieces = p.n_slabs * (kSlabChunks * BN * 2);
    for (int i = tid; i < pieces; i += kThreads) wdst[i] = wsrc[i];
  }
  __syncthreads();
  const int frag_off = (lane >> 5) * (BN * 8) + (lane & 31) * 8;  // halfs
  const int ohow = p.OH * p.OW;
  const int n_full = p.n_chunks / kSlabChunks;
  const int rem = p.n_chunks - n_full * kSlabChunks;
  const int n_wave_tiles = (p.M + 32 * PT - 1) / (32 * PT);
  for (int wt = seq * WAVES + wave; wt < n_wave_tiles; wt += n_seq * WAVES) {
    int n0 = 0;
    unsigned base[PT];
    int pn[PT], poh[PT], pow_[PT];
    bool mvalid[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const int m = (wt * PT + pt) * 32 + (lane & 31);
      int n, pix, oh, ow;
      mvalid[pt] = m < p.M;
      divmod_small(mvalid[pt] ? m : 0, ohow, p.rcp_ohow, n, pix);
      divmod_small(pix, p.OW, p.rcp_ow, oh, ow);
      const int iy = oh * p.stride - p.pad_h + p.ig.halo;
      const int ix = ow * p.stride - p.pad_w + p.ig.halo;
      if (pt == 0) n0 = __builtin_amdgcn_readfirstlane(n);
      base[pt] = mvalid[pt]
                     ? static_cast<unsigned>(((((n - n0) * p.ig.groups + (lane >> 5)) * p.ig.hp + iy) *
                                                  p.ig.wp + ix) * 16)
                     : 0x80000000u;
      pn[pt] = n;
      poh[pt] = oh;
      pow_[pt] = ow;
    }
    const size_t in_off = static_cast<size_t>(n0) * p.img_bytes;
    const size_t in_left = p.in_bytes - in_off;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.in) + in_off), 0,
        static_cast<unsigned>(in_left < 0x7fffffffu ? in_left : 0x7fffffffu), 0x00020000);
    uint4_t xf[kPrefetch][PT];
    float16_t acc[NB][PT];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][pt][i] = 0.f;
    ChunkWalk walk{0, 0, 0u, 0u};
#pragma unroll
    for (int d = 0; d < kPrefetch; ++d) {
      const unsigned soff = walk.off();
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        xf[d][pt] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base[pt], soff, 0);
      }
      walk.advance(p);
    }
    for (int s = 0; s < n_full; ++s) {
      conv_slab<NB, PT, kSlabChunks>(p, rsrc, smem + s * SLAB_HALFS + frag_off, walk, base, xf, acc);
    }
    if (rem) {
      const _Float16* wslab = smem + n_full * SLAB_HALFS + frag_off;
      conv_slab<NB, PT, 4>(p, rsrc, wslab, walk, base, xf, acc);
      if (rem > 4) conv_slab<NB, PT, 4, 4>(p, rsrc, wslab + 4 * BN * kChunk, walk, base, xf, acc);
    }
    conv_epilogue<NB, PT>(acc, p, n_tile, pn, poh, pow_, mvalid, lane);
  }
}

// conv_resident_kernel with the 3x3 / stride-2 max-pool that follows the convolution taken INSIDE
// (stem: 3x3 80->192 -> max-pool -> mixed0).  The conv tensor (21 x 51 x 192 per example, 0.4 MB
// written and read back: the worst launch of round 3 was the pooled re-read) never exists; only
// the pooled tensor (10 x 25 x 192) is stored, and mixed0's heads become an ordinary grouped 1x1.
//
// Max-pooling commutes with every monotone map, so pooling the fp16 results of shift + ReLU is the
// same as Keras' order -- results are bit-identical to the separate max-pool.
//
//  * A wave owns 32 POSITIONS of the flattened (example, conv column) index -- fragment f covers
//    positions 30 f .. 30 f + 31, an overlap of two so that every 3-wide window that starts in
//    the first 30 lies inside the wave (6 % of the MFMA work is recomputed) -- and walks DOWN the
//    map two conv rows per step (the two pixel fragments of conv_slab<NB,2>: same K order, same
//    weights resident in LDS, same straight-line slab code as conv_resident_kernel).
//  * Vertical maximum: rows 2i, 2i+1, 2i+2 of one position sit in the same lane of three
//    accumulator sets, so it is register-wise (v_pk_max_f16); max(row 2i+2, row 2i+3) is carried
//    into the next step as 8 packed dwords per 32-cout subtile.
//  * Horizontal maximum: positions +1 / +2 are lanes +1 / +2 of the same register
//    (ds_bpermute_b32, no LDS memory); lanes on even columns then hold pooled pixels and store
//    16-byte pieces after the usual permlane32_swap pairing.
//  * The last step has one conv row only (row 2 PH) and runs conv_slab<NB,1>.
template <int NB, int PT>
__device__ __forceinline__ void resident_rows(const ConvArgs& p, const __amdgpu_buffer_rsrc_t rsrc,
                                              const _Float16* wfrag, const unsigned (&base)[PT],
                                              float16_t (&acc)[NB][PT]) {
  constexpr int BN = NB * 32;
  constexpr int SLAB_HALFS = kSlabChunks * BN * kChunk;
  constexpr int kPrefetch = prefetch_depth(NB, PT);
  const int n_full = p.n_chunks / kSlabChunks;
  const int rem = p.n_chunks - n_full * kSlabChunks;
  uint4_t xf[kPrefetch][PT];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nb][pt][i] = 0.f;
  ChunkWalk walk{0, 0, 0u, 0u};
#pragma unroll
  for (int d = 0; d < kPrefetch; ++d) {
    const unsigned soff = walk.off();
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) xf[d][pt] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base[pt], soff, 0);
    walk.advance(p);
  }
  for (int s = 0; s < n_full; ++s) {
    conv_slab<NB, PT, kSlabChunks>(p, rsrc, wfrag + s * SLAB_HALFS, walk, base, xf, acc);
  }
  if (rem) {
    const _Float16* wslab = wfrag + n_full * SLAB_HALFS;
    conv_slab<NB, PT, 4>(p, rsrc, wslab, walk, base, xf, acc);
    if (rem > 4) conv_slab<NB, PT, 4, 4>(p, rsrc, wslab + 4 * BN * kChunk, walk, base, xf, acc);
  }
}

// shift + ReLU + fp16 of one conv row held in accumulators: pk[nb][q][hq] = couts
// nb*32 + 8q + 4*(lane>>5) + 2hq + {0,1} of the lane's position (conv_epilogue's packing)
template <int NB>
__device__ __forceinline__ void pack_row(const float16_t (&acc)[NB], const ConvArgs& p, int n_tile, int hi,
                                         unsigned (&pk)[NB][4][2]) {
  const ConvBranch& b = p.br[0];
  const half2_t zero2 = {static_cast<_Float16>(0.f), static_cast<_Float16>(0.f)};
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int cbase = (n_tile * NB + nb) * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      typedef float f4_t __attribute__((ext_vector_type(4)));
      typedef const f4_t __attribute__((address_space(4))) * const_f4_ptr;
      const_f4_ptr sp = (const_f4_ptr)(reinterpret_cast<uintptr_t>(b.shift + (cbase + 8 * q)));
      const f4_t l4 = sp[0], u4 = sp[1];   // the shift array is padded past Cout
      const float2_t s0 = hi ? float2_t{u4[0], u4[1]} : float2_t{l4[0], l4[1]};
      const float2_t s1 = hi ? float2_t{u4[2], u4[3]} : float2_t{l4[2], l4[3]};
      const float2_t v0 = float2_t{acc[nb][4 * q], acc[nb][4 * q + 1]} + s0;
      const float2_t v1 = float2_t{acc[nb][4 * q + 2], acc[nb][4 * q + 3]} + s1;
      pk[nb][q][0] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_convertvector(v0, half2_t), zero2));
      pk[nb][q][1] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_convertvector(v1, half2_t), zero2));
    }
  }
}

__device__ __forceinline__ unsigned pk_max(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(half2_t, a),
                                                                __builtin_bit_cast(half2_t, b)));
}

template <int NB>
__global__ __launch_bounds__(512, 1) void conv_pool_resident_kernel(ConvArgs p) {
  constexpr int BN = NB * 32;
  constexpr int WAVES = 8, kThreads = WAVES * 64;
  constexpr int kNew = 30;   // positions a fragment owns (the last two belong to the next one)
  extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int blk = blockIdx.x;
  const int n_tile = (blk >> 3) % p.n_tiles;
  const int seq = (blk & 7) + 8 * ((blk >> 3) / p.n_tiles);
  const int n_seq = 8 * ((static_cast<int>(gridDim.x) >> 3) / p.n_tiles);
  {
    const uint4_t* wsrc = reinterpret_cast<const uint4_t*>(p.w) +
                          static_cast<size_t>(n_tile) * p.n_slabs * (kSlabChunks * BN * 2);
    uint4_t* wdst = reinterpret_cast<uint4_t*>(smem);
    const int pieces = p.n_slabs * (kSlabChunks * BN * 2);
    for (int i = tid; i < pieces; i += kThreads) wdst[i] = wsrc[i];
  }
  __syncthreads();
  const _Float16* wfrag = smem + (lane >> 5) * (BN * 8) + (lane & 31) * 8;
  const ConvBranch& b = p.br[0];
  const int PH = b.og.h, PW = b.og.w;          // pooled map
  const int n_pos = p.N * p.OW;
  const int n_frag = (n_pos + kNew - 1) / kNew;
  const unsigned row_b = static_cast<unsigned>(p.ig.wp) * 16u;
  const int hi = lane >> 5, l32 = lane & 31;
  const unsigned gstride = static_cast<unsigned>(b.og.hp * b.og.wp);
  uint4_t* outp = reinterpret_cast<uint4_t*>(b.out);
  for (int f = seq * WAVES + wave; f < n_frag; f += n_seq * WAVES) {
    const int pos = f * kNew + l32;
    const bool valid = pos < n_pos;
    int n, col;
    divmod_small(valid ? pos : 0, p.OW, p.rcp_ow, n, col);
    const int n0 = __builtin_amdgcn_readfirstlane(n);
    // (example n - n0, channel group lane>>5, input row halo, input column col + halo): 'valid' conv
    const unsigned base0 =
        valid ? static_cast<unsigned>(((((n - n0) * p.ig.groups + hi) * p.ig.hp + p.ig.halo) * p.ig.wp + col +
                                       p.ig.halo) * 16)
              : 0x80000000u;
    const size_t in_off = static_cast<size_t>(n0) * p.img_bytes;
    const size_t in_left = p.in_bytes - in_off;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.in) + in_off), 0,
        static_cast<unsigned>(in_left < 0x7fffffffu ? in_left : 0x7fffffffu), 0x00020000);
    // pooled pixel (n, s - 1, col / 2) leaves from this lane at step s
    const bool emits = valid && l32 < kNew && (col & 1) == 0 && (col >> 1) < PW;
    const unsigned obase0 = static_cast<unsigned>(
        ((n * b.og.groups + b.out_goff) * b.og.hp + b.og.halo) * b.og.wp + (col >> 1) + b.og.halo);
    const int nb_addr1 = ((lane + 1) & 63) * 4, nb_addr2 = ((lane + 2) & 63) * 4;
    unsigned carry[NB][4][2];   // max(row 2s, row 2s+1) of the previous step
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int q = 0; q < 4; ++q) carry[nb][q][0] = carry[nb][q][1] = 0u;
    // Blank-row skipping (ConvArgs::blank_row = first blank-determined POOLED row per example): the wave walks
    // down only as far as the deepest pile-up among its (at most two) examples reaches -- pooled rows
    // 0 .. s_end - 1 -- and copies the rows below from the all-blank image's response (the values the walk would
    // produce there, bit for bit).  A lane whose own example turns blank earlier computes its blank rows: same bits.
    int s_end = PH;
    if (p.blank_row != nullptr) {
      int mine = valid ? min(p.blank_row[n], PH) : 0;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) mine = max(mine, __shfl_xor(mine, off));
      s_end = __builtin_amdgcn_readfirstlane(mine);
    }
    for (int s = 0; s <= s_end; ++s) {
      if (s_end == 0) break;      // every pooled row of this fragment is blank-determined
      unsigned top[NB][4][2];   // conv row 2s
      if (s < s_end) {
        unsigned base[2] = {base0, base0};
        if (valid) {
          base[0] = base0 + static_cast<unsigned>(2 * s) * row_b;
          base[1] = base[0] + row_b;
        }
        float16_t acc[NB][2];
        resident_rows<NB, 2>(p, rsrc, wfrag, base, acc);
        float16_t r0[NB], r1[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          r0[nb] = acc[nb][0];
          r1[nb] = acc[nb][1];
        }
        unsigned bot[NB][4][2];
        pack_row<NB>(r0, p, n_tile, hi, top);
        pack_row<NB>(r1, p, n_tile, hi, bot);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
              const unsigned t = top[nb][q][hq];
              top[nb][q][hq] = pk_max(carry[nb][q][hq], t);   // rows 2s-2, 2s-1, 2s (s = 0: row 0 alone, unused)
              carry[nb][q][hq] = pk_max(t, bot[nb][q][hq]);
            }
      } else {   // the last pooled row's third conv row
        unsigned base[1] = {valid ? base0 + static_cast<unsigned>(2 * s) * row_b : base0};
        float16_t acc[NB][1];
        resident_rows<NB, 1>(p, rsrc, wfrag, base, acc);
        float16_t r0[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) r0[nb] = acc[nb][0];
        pack_row<NB>(r0, p, n_tile, hi, top);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) top[nb][q][hq] = pk_max(carry[nb][q][hq], top[nb][q][hq]);
      }
      if (s == 0) continue;   // wave-uniform
      const unsigned obase = obase0 + static_cast<unsigned>(s - 1) * static_cast<unsigned>(b.og.wp);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int cbase = (n_tile * NB + nb) * 32;
        unsigned hm[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int hq = 0; hq < 2; ++hq) {
            const unsigned v = top[nb][q][hq];
            const unsigned v1 = static_cast<unsigned>(__builtin_amdgcn_ds_bpermute(nb_addr1, static_cast<int>(v)));
            const unsigned v2 = static_cast<unsigned>(__builtin_amdgcn_ds_bpermute(nb_addr2, static_cast<int>(v)));
            hm[q][hq] = pk_max(pk_max(v, v1), v2);
          }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const auto d0 = __builtin_amdgcn_permlane32_swap(hm[2 * t][0], hm[2 * t + 1][0], false, false);
          const auto d1 = __builtin_amdgcn_permlane32_swap(hm[2 * t][1], hm[2 * t + 1][1], false, false);
          const uint4_t piece = {d0[0], d1[0], d0[1], d1[1]};
          const int group = cbase / 8 + 2 * t + hi;
          if (emits && group * 8 < b.Cout) outp[obase + static_cast<unsigned>(group) * gstride] = piece;
        }
      }
    }
    if (s_end < PH && emits) {   // the blank-determined pooled rows of this lane's column: copied
      const uint4_t* src = reinterpret_cast<const uint4_t*>(p.blank_src);
      const unsigned rel0 = static_cast<unsigned>(b.og.halo * b.og.wp + (col >> 1) + b.og.halo);
      const unsigned img = static_cast<unsigned>(n * b.og.groups) * gstride;
      for (int pr = s_end; pr < PH; ++pr) {
        const unsigned rel = rel0 + static_cast<unsigned>(pr * b.og.wp);
#pragma unroll
        for (int j = 0; j < NB * 2; ++j) {   // lanes l / l + 32 take alternate 8-cout groups of this tile
          const int group = n_tile * NB * 4 + 2 * j + hi;
          if (group * 8 < b.Cout) {
            const unsigned at = static_cast<unsigned>(b.out_goff + group) * gstride + rel;
            outp[img + at] = src[at];
          }
        }
      }
    }
  }
}

// MaxPooling2D(3, strides 2) fused into the 1x1 convolution that consumes it (stem:
// maxpool -> 64->80): the pooled tensor (0.16 MB / example written and read back) never
// exists.  A pixel fragment is the element-wise maximum of the nine 16-byte pieces of its
// window, taken in registers (v_pk_max_f16) -- exact, so results are bit-identical to the
// separate max-pool kernel.  K is short (Cin/16 chunks): the whole weight tile sits in LDS,
// fragments of chunk c+1 are in flight while chunk c multiplies.  One 32-pixel fragment
// per wave (PT = 1) keeps the 2 x 9 outstanding loads within the register budget.
template <int NB, int WAVES = 4>
__global__ __launch_bounds__(WAVES * 64, WAVES == 4 ? 2 : 1) void conv_pool1x1_kernel(ConvArgs p) {
  constexpr int BN = NB * 32;
  constexpr int PT = 1;
  constexpr int kThreads = WAVES * 64;
  extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int nwg = gridDim.x;
  const int xq = nwg >> 3, xr = nwg & 7;
  const int xcd = blockIdx.x & 7, xi = blockIdx.x >> 3;
  const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
  const int n_tile = logical % p.n_tiles;
  const int m_block = (logical / p.n_tiles) * (WAVES * 32);

  {  // whole weight tile -> LDS
    const int pieces = p.n_slabs * kSlabChunks * BN * 2;
    const uint4_t* src = reinterpret_cast<const uint4_t*>(p.w) + static_cast<size_t>(n_tile) * pieces;
    uint4_t* dst = reinterpret_cast<uint4_t*>(smem);
    for (int i = tid; i < pieces; i += kThreads) dst[i] = src[i];
  }
  int pn[PT], poh[PT], pow_[PT];
  bool mvalid[PT];
  const int m = m_block + wave * 32 + (lane & 31);
  mvalid[0] = m < p.M;
  int n, pix, oh, ow;
  divmod_small(mvalid[0] ? m : 0, p.OH * p.OW, p.rcp_ohow, n, pix);
  divmod_small(pix, p.OW, p.rcp_ow, oh, ow);
  pn[0] = n;
  poh[0] = oh;
  pow_[0] = ow;
  const int n0 = __builtin_amdgcn_readfirstlane(n);
  // window origin: pooled pixel (oh, ow) covers input rows 2oh..2oh+2, cols 2ow..2ow+2
  const unsigned base =
      mvalid[0] ? static_cast<unsigned>(((((n - n0) * p.ig.groups + (lane >> 5)) * p.ig.hp + 2 * oh +
                                          p.ig.halo) * p.ig.wp + 2 * ow + p.ig.halo) * 16)
                : 0x80000000u;
  const size_t in_off = static_cast<size_t>(n0) * p.img_bytes;
  const size_t in_left = p.in_bytes - in_off;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.in) + in_off), 0,
      static_cast<unsigned>(in_left < 0x7fffffffu ? in_left : 0x7fffffffu), 0x00020000);
  const unsigned row_b = static_cast<unsigned>(p.ig.wp) * 16u;

  float16_t acc[NB][PT];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nb][0][i] = 0.f;

  half8_t win[2][9];
#define DV_LOAD_WINDOW(slot_, cc_)                                                          \
  {                                                                                         \
    const unsigned so_ = static_cast<unsigned>(cc_) * p.chunk_stride;                       \
    _Pragma("unroll") for (int t_ = 0; t_ < 9; ++t_) win[slot_][t_] = __builtin_bit_cast(  \
        half8_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, base,                          \
                                                       so_ + (t_ / 3) * row_b + (t_ % 3) * 16, 0)); \
  }
  DV_LOAD_WINDOW(0, 0)
  __syncthreads();
  const _Float16* wfrag = smem + (lane >> 5) * (BN * 8) + (lane & 31) * 8;
  for (int cc = 0; cc < p.n_chunks; cc += 2) {   // two chunks per trip: static window slots
    if (cc + 1 < p.n_chunks) DV_LOAD_WINDOW(1, cc + 1)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (half == 1) {
        if (cc + 1 >= p.n_chunks) break;
        if (cc + 2 < p.n_chunks) DV_LOAD_WINDOW(0, cc + 2)
      }
      half8_t x = win[half][0];
#pragma unroll
      for (int t = 1; t < 9; ++t) x = __builtin_elementwise_max(x, win[half][t]);
      const _Float16* wc = wfrag + (cc + half) * (BN * kChunk);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const half8_t wf = *reinterpret_cast<const half8_t*>(wc + nb * 32 * 8);
        acc[nb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, x, acc[nb][0], 0, 0, 0);
      }
    }
  }
#undef DV_LOAD_WINDOW
  conv_epilogue<NB, PT>(acc, p, n_tile, pn, poh, pow_, mvalid, lane);
}

template <int NB>
constexpr size_t conv_lds_bytes() {
  return static_cast<size_t>(2) * kSlabChunks * NB * 32 * kChunk * 2;
}

// First convolution fused with preprocess_images: reads the uint8 HWC pileup
// tensor the encoder wrote (C <= 8 channels), normalises (x-128)/128 in
// registers and multiplies on MFMA.  K layout: one 16-wide chunk = two filter
// taps x 8 "channels" (C real + zero-weight padding), so a 3x3x7 filter is 5
// chunks instead of the 9 half-empty ones of a 16-channel padded fp16 image,
// and the 0.7 MB/example fp16 staging tensor disappears (HBM: 155 KB read
// instead of 155 KB read + 707 KB written + 707 KB read).
// A lane's fragment = the 8 bytes at (pixel, tap) -- unaligned, fetched as the
// 3 aligned dwords around it and funnel-shifted; byte C..7 belong to the next
// pixel and meet zero weights.  'valid' convolutions only, Cout <= 32.
// The caller's two pointers (uint8 images in, probabilities out) are read by the kernels from
// this device-side table instead of being kernel arguments: a captured forward then depends on
// the batch size only, and a caller that hands over a fresh tensor per region replays the same
// hipGraph (dv_model_infer writes the table with a one-thread kernel ahead of every forward).
struct ExtPtrs {
  const uint8_t* images;
  float* probs;
  const int32_t* rows_hint;   // dv_model_infer_rows: per image, rows at or below rows_hint[i] + rows_add are all zero
  int rows_add;
};

__global__ void set_ext_kernel(ExtPtrs* ext, const uint8_t* images, float* probs, const int32_t* rows_hint, int rows_add) {
  ext->images = images;
  ext->probs = probs;
  ext->rows_hint = rows_hint;
  ext->rows_add = rows_add;
}

__device__ __forceinline__ const uint8_t* ext_images(const ExtPtrs* ext, size_t off) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(ext->images + off);
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
  const unsigned up = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
  return reinterpret_cast<const uint8_t*>((static_cast<unsigned long long>(up) << 32) | lo);
}

struct FirstConvArgs {
  const ExtPtrs* ext;       // images = ext->images + in_off: [N][H][W][C]
  size_t in_off;
  const _Float16* w;        // packed [chunk][2 k-groups = taps][32][8]
  const float* shift;
  _Float16* out;
  TensorGeom og;
  int N, H, W, C, Cout;
  int OH, OW, KH, KW, stride;
  int M, n_chunks;
  unsigned in_bytes;
  float rcp_ow, rcp_ohow;
  // C in 9..16 (the long-read channel sets: 9 = ONT_R104, 10 = PACBIO): a K chunk is ONE tap x 16
  // "channels" -- k-group 0 = bytes 0..7 of the pixel, k-group 1 = bytes 8..15 (bytes C.. belong to the
  // next pixel and meet zero weights) -- instead of two taps x 8
  int wide;
};

constexpr int kFirstMaxChunks = 13;  // up to 5x5 taps (C <= 8) / 3x3 taps (C <= 16)

template <int PT, int UNROLL = kFirstUnroll>
__global__ __launch_bounds__(kConvThreads) void conv_first_u8_kernel(FirstConvArgs p) {
  __shared__ __attribute__((aligned(16))) _Float16 wl[kFirstMaxChunks * 32 * kChunk];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int hi = lane >> 5;
  const int m_block = blockIdx.x * (128 * PT);

  {  // all weights (<= 13 KB) -> LDS
    const uint4_t* src = reinterpret_cast<const uint4_t*>(p.w);
    uint4_t* dst = reinterpret_cast<uint4_t*>(wl);
    for (int i = tid; i < p.n_chunks * 64; i += kConvThreads) dst[i] = src[i];
  }
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(ext_images(p.ext, p.in_off)), 0, p.in_bytes, 0x00020000);

  unsigned base[PT], obase[PT];
  bool mvalid[PT];
  const int ohow = p.OH * p.OW;
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int m = m_block + (wave * PT + pt) * 32 + (lane & 31);
    mvalid[pt] = m < p.M;
    int n, pix, oh, ow;
    divmod_small(mvalid[pt] ? m : 0, ohow, p.rcp_ohow, n, pix);
    divmod_small(pix, p.OW, p.rcp_ow, oh, ow);
    base[pt] = mvalid[pt] ? static_cast<unsigned>(((n * p.H + oh * p.stride) * p.W +
                                                   ow * p.stride) * p.C)
                          : 0x80000000u;
    obase[pt] = static_cast<unsigned>((n * p.og.groups * p.og.hp + oh + p.og.halo) * p.og.wp +
                                      ow + p.og.halo);
  }
  const int taps = p.KH * p.KW;
  float16_t acc[PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[pt][i] = 0.f;
  __syncthreads();

  // uint8 -> fp16 without per-byte converts: 0x6400 | b is the fp16 number 1024 + b, and
  // (1024 + b) * 2^-7 - 9 = (b - 128) / 128 exactly -- one v_perm_b32 and one packed FMA
  // per two channels.
  auto normalise = [](unsigned lo, unsigned up) {
    const half2_t scale = {static_cast<_Float16>(0.0078125f), static_cast<_Float16>(0.0078125f)};
    const half2_t bias = {static_cast<_Float16>(-9.0f), static_cast<_Float16>(-9.0f)};
    const unsigned k = 0x64646464u;
    const half2_t h01 = __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(lo, k, 0x00050004u));
    const half2_t h23 = __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(lo, k, 0x00070006u));
    const half2_t h45 = __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(up, k, 0x00050004u));
    const half2_t h67 = __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(up, k, 0x00070006u));
    const half2_t a = h01 * scale + bias, b = h23 * scale + bias;
    const half2_t c = h45 * scale + bias, d = h67 * scale + bias;
    return half8_t{a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
  };
  typedef unsigned uint3_t __attribute__((ext_vector_type(3)));
  if (p.n_chunks <= UNROLL) {
    // every fragment of the tile is requested before the first one is used
    uint3_t d[UNROLL][PT];
    unsigned sh[UNROLL][PT];
#pragma unroll
    for (int kc = 0; kc < UNROLL; ++kc) {
      const int t = min(p.wide ? kc : 2 * kc + hi, taps - 1);
      const int kh = t / p.KW, kw = t - kh * p.KW;
      const unsigned toff = static_cast<unsigned>((kh * p.W + kw) * p.C + (p.wide ? 8 * hi : 0));
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const unsigned a = base[pt] + toff;
        sh[kc][pt] = (a & 3u) * 8u;
        d[kc][pt] = kc < p.n_chunks ? __builtin_amdgcn_raw_buffer_load_b96(rsrc, a & ~3u, 0, 0)
                                    : uint3_t{0u, 0u, 0u};
      }
    }
#pragma unroll
    for (int kc = 0; kc < UNROLL; ++kc) {
      if (kc < p.n_chunks) {
        const half8_t wf = *reinterpret_cast<const half8_t*>(
            wl + kc * 32 * kChunk + hi * (32 * 8) + (lane & 31) * 8);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          const unsigned lo = __builtin_amdgcn_alignbit(d[kc][pt][1], d[kc][pt][0], sh[kc][pt]);
          const unsigned up = __builtin_amdgcn_alignbit(d[kc][pt][2], d[kc][pt][1], sh[kc][pt]);
          acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, normalise(lo, up), acc[pt], 0, 0, 0);
        }
      }
    }
  } else {
    for (int kc = 0; kc < p.n_chunks; ++kc) {
      // this lane-half's tap (wide: its half of the tap's channels); past the last tap the weights are zero
      const int t = min(p.wide ? kc : 2 * kc + hi, taps - 1);
      const int kh = t / p.KW, kw = t - kh * p.KW;
      const unsigned toff = static_cast<unsigned>((kh * p.W + kw) * p.C + (p.wide ? 8 * hi : 0));
      const half8_t wf = *reinterpret_cast<const half8_t*>(
          wl + kc * 32 * kChunk + hi * (32 * 8) + (lane & 31) * 8);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const unsigned a = base[pt] + toff;
        const uint3_t dd = __builtin_amdgcn_raw_buffer_load_b96(rsrc, a & ~3u, 0, 0);
        const unsigned s8 = (a & 3u) * 8u;
        const unsigned lo = __builtin_amdgcn_alignbit(dd[1], dd[0], s8);
        const unsigned up = __builtin_amdgcn_alignbit(dd[2], dd[1], s8);
        acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, normalise(lo, up), acc[pt], 0, 0, 0);
      }
    }
  }

  // epilogue (same piece pairing as conv_mfma_kernel, one 32-cout tile)
  const unsigned gstride = static_cast<unsigned>(p.og.hp * p.og.wp);
  uint4_t* outp = reinterpret_cast<uint4_t*>(p.out);
  const half2_t zero2 = {static_cast<_Float16>(0.f), static_cast<_Float16>(0.f)};
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    unsigned pk[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = 8 * q + 4 * hi;
#pragma unroll
      for (int hq = 0; hq < 2; ++hq) {
        const float2_t sv = {co + 2 * hq < p.Cout ? p.shift[co + 2 * hq] : 0.f,
                             co + 2 * hq + 1 < p.Cout ? p.shift[co + 2 * hq + 1] : 0.f};
        const float2_t v = float2_t{acc[pt][4 * q + 2 * hq], acc[pt][4 * q + 2 * hq + 1]} + sv;
        half2_t h = __builtin_convertvector(v, half2_t);
        h = __builtin_elementwise_max(h, zero2);
        pk[q][hq] = __builtin_bit_cast(unsigned, h);
      }
    }
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      const auto d0 = __builtin_amdgcn_permlane32_swap(pk[2 * t2][0], pk[2 * t2 + 1][0], false, false);
      const auto d1 = __builtin_amdgcn_permlane32_swap(pk[2 * t2][1], pk[2 * t2 + 1][1], false, false);
      const uint4_t piece = {d0[0], d1[0], d0[1], d1[1]};
      const int group = 2 * t2 + hi;
      if (mvalid[pt] && group * 8 < p.Cout) {
        outp[obase[pt] + static_cast<unsigned>(group) * gstride] = piece;
      }
    }
  }
}

// uint8 [N,H,W,C] -> fp16 C8 [N][2][hp][wp][8]: (x - 128) / 128, exact in fp16.
__global__ void preprocess_kernel(const ExtPtrs* ext, size_t in_off, _Float16* out, size_t n_pix, int C,
                                  int H, int W, TensorGeom og) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i >= n_pix) return;
  const uint8_t* in = ext->images + in_off;
  const uint8_t* px = in + i * C;
  _Float16 v[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    v[c] = c < C ? static_cast<_Float16>((static_cast<float>(px[c]) - 128.0f) / 128.0f)
                 : static_cast<_Float16>(0.f);
  }
  const size_t n = i / (static_cast<size_t>(H) * W);
  const int pix = static_cast<int>(i - n * (static_cast<size_t>(H) * W));
  const int y = pix / W, x = pix - y * W;
  uint4* dst = reinterpret_cast<uint4*>(out);
  const size_t plane = static_cast<size_t>(og.hp) * og.wp;
  const size_t at = (n * 2) * plane + static_cast<size_t>(y + og.halo) * og.wp + x + og.halo;
  dst[at] = *reinterpret_cast<uint4*>(&v[0]);
  dst[at + plane] = *reinterpret_cast<uint4*>(&v[8]);
}

// Blank-row skipping (opt-in, DV_BLANK_SKIP): one workgroup per image finds the last row that
// holds a nonzero byte, scanning from the bottom (a 30x pileup is zero below row ~40, so about
// 60 % of the image is read once), and turns it into the first blank-determined row of the
// stem's tensors: an output whose receptive field sees only zero rows equals the all-blank
// image's output at the same position.
//   conv1 3x3/2 valid: rows 2y..2y+2   -> y >= ceil(r / 2)         (= conv2, 3x3 valid on those)
//   conv3 3x3 same:    rows y-1..y+1   -> y >= t2 + 1
//   max-pool 3x3/2:    rows 2y..2y+2   -> y >= ceil(t3 / 2)        (= the 1x1 and the 3x3 valid 80->192)
// thr[k * stride + n], k = 0: rows used, 1: conv2 output, 2: stem_b output, 3: 3x3 80->192 output, 4: the same, pooled.
// What the consumers of the skipping kernels read (thr rows 5 and 6): blank tiles beyond these rows are not even
// copied.  need2 = conv2 rows under stem_b's computed tiles (pooled tiles of kStemB_PH rows starting above t4: pooled
// row py reads conv3 rows 2py..2py+2, conv3 row y conv2 rows y-1..y+1); need4 = stem_b rows under the 3x3 80->192's
// walk, which goes down to the deepest pooled threshold among the examples a 32-position fragment spans (conv row r
// reads rows r..r+2; the walk's last row is 2 s_end).
// (stem_b_fused / conv4_walks = 0: the consumer is a per-layer kernel that may read every row -- everything is wanted.)
__global__ void blank_need_kernel(int* thr, int stride, int n, int oh2, int ph_b, int ow4, int p4, int stem_b_fused,
                                  int conv4_walks) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int t4 = min(thr[2 * stride + i], ph_b);
  const int kb = (t4 + dv::kStemB_PH - 1) / dv::kStemB_PH;
  thr[5 * stride + i] = !stem_b_fused ? oh2 : kb > 0 ? min(oh2, 2 * dv::kStemB_PH * kb + 2) : 0;
  const int span = 31 / max(ow4, 1) + 1;
  int m5 = 0;
  for (int j = max(0, i - span); j <= min(n - 1, i + span); ++j) m5 = max(m5, min(thr[4 * stride + j], p4));
  thr[6 * stride + i] = !conv4_walks ? ph_b : m5 > 0 ? min(ph_b, 2 * m5 + 3) : 0;
}

__global__ __launch_bounds__(256) void blank_rows_kernel(const ExtPtrs* ext, size_t in_off, int n_hint0, int H,
                                                         int row_bytes, int* thr, int stride) {
  __shared__ int last;
  const uint8_t* images = ext->images + in_off;
  const int n = blockIdx.x, tid = threadIdx.x;
  const unsigned img_bytes = static_cast<unsigned>(H) * row_bytes;   // multiple of 4 (checked by the host)
  const uint32_t* img = reinterpret_cast<const uint32_t*>(images + static_cast<size_t>(n) * img_bytes);
  const int n_dw = static_cast<int>(img_bytes / 4);
  if (tid == 0) last = -1;
  __syncthreads();
  constexpr int kPer = 16;   // dwords per thread and trip: 16 KB of the image per barrier
  // dv_model_infer_rows: the caller (the encoder that drew the images) states the rows used -- no scan
  const bool hinted = ext->rows_hint != nullptr;   // uniform
  const int hinted_rows = hinted ? max(0, min(H, ext->rows_hint[n_hint0 + n] + ext->rows_add)) : 0;
  for (int hi = hinted ? 0 : n_dw; hi > 0; hi -= 256 * kPer) {
    uint32_t v[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {   // all loads in flight before the first compare
      const int i = hi - 1 - (k * 256 + tid);
      v[k] = i >= 0 ? img[i] : 0u;
    }
    int mine = -1;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int i = hi - 1 - (k * 256 + tid);
      if (v[k] != 0) mine = max(mine, 4 * i + 3 - (__clz(v[k]) >> 3));   // its highest nonzero byte
    }
    if (mine >= 0) atomicMax(&last, mine);
    __syncthreads();
    if (last >= 0) break;   // uniform: read after the barrier
  }
  if (tid == 0) {
    const int r = hinted ? hinted_rows : last < 0 ? 0 : last / row_bytes + 1;
    const int t2 = (r + 1) / 2, t3 = t2 + 1, t4 = (t3 + 1) / 2;
    thr[n] = r;
    thr[stride + n] = t2;
    thr[2 * stride + n] = t4;
    thr[3 * stride + n] = t4;
    thr[4 * stride + n] = (t4 + 1) / 2;   // the 3x3 80->192's output max-pooled (3x3 / 2) inside its producer: rows 2s .. 2s+2
  }
}

struct PoolArgs {
  const _Float16* in;     // maxpool3s2_kernel: fp16
  const float* in32;      // avgpool3s1_kernel: the float32 raw projection
  _Float16* out;
  float* out32;           // avgpool3s1_kernel: non-NULL = the pooled tensor is float32 (the last block's, read by the head)
  int lo_in_groups;       // maxpool3s2_kernel: > 0 = the input is wide (hi groups, then lo groups: precise mode)
  int lo_out_groups;      // > 0 = the output tensor is wide: the lo pieces go lo_out_groups channel groups further
  TensorGeom ig, og;
  int N, C, OH, OW;
  int out_goff;
  const float* shift;  // avgpool only: per-channel shift + ReLU after the average, or NULL
};

// MaxPooling2D(3, strides=2, 'valid'), C8 layout; one thread = one 16-byte piece.  Wide tensors (precise mode): the
// maximum of hi + lo is the lexicographic maximum of (hi, lo) -- |lo| is at most half an ulp of hi.
__global__ void maxpool3s2_kernel(PoolArgs p) {
  const int cg = p.C / 8;
  const size_t total = static_cast<size_t>(p.N) * cg * p.OH * p.OW;
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i >= total) return;
  const int ow = i % p.OW;
  size_t t = i / p.OW;
  const int oh = t % p.OH;
  t /= p.OH;
  const int g = t % cg;
  const int n = t / cg;
  const size_t plane = static_cast<size_t>(p.ig.hp) * p.ig.wp;
  const half8_t* src = reinterpret_cast<const half8_t*>(p.in) + (static_cast<size_t>(n) * p.ig.groups + g) * plane;
  const half8_t* src_lo = src + static_cast<size_t>(p.lo_in_groups) * plane;
  half8_t best, best_lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    best[j] = static_cast<_Float16>(-65504.f);
    best_lo[j] = static_cast<_Float16>(0.f);
  }
  const size_t at0 = static_cast<size_t>(oh * 2 + p.ig.halo) * p.ig.wp + ow * 2 + p.ig.halo;
  if (p.lo_in_groups > 0) {   // uniform; all eighteen pieces in flight before the first compare
    half8_t v[9], l[9];
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      const size_t at = at0 + static_cast<size_t>(t9 / 3) * p.ig.wp + t9 % 3;
      v[t9] = src[at];
      l[t9] = src_lo[at];
    }
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool take = v[t9][j] > best[j] || (v[t9][j] == best[j] && l[t9][j] > best_lo[j]);
        best[j] = take ? v[t9][j] : best[j];
        best_lo[j] = take ? l[t9][j] : best_lo[j];
      }
    }
  } else {
    for (int dh = 0; dh < 3; ++dh)
      for (int dw = 0; dw < 3; ++dw) {
        const half8_t v = src[at0 + static_cast<size_t>(dh) * p.ig.wp + dw];
#pragma unroll
        for (int j = 0; j < 8; ++j) best[j] = v[j] > best[j] ? v[j] : best[j];
      }
  }
  const size_t oplane = static_cast<size_t>(p.og.hp) * p.og.wp;
  half8_t* dst = reinterpret_cast<half8_t*>(p.out) + (static_cast<size_t>(n) * p.og.groups + p.out_goff + g) * oplane +
                 static_cast<size_t>(oh + p.og.halo) * p.og.wp + ow + p.og.halo;
  *dst = best;
  if (p.lo_out_groups > 0) dst[static_cast<size_t>(p.lo_out_groups) * oplane] = best_lo;
}

// AveragePooling2D(3, strides=1, 'same'): divisor = number of valid cells.  The input is the float32 raw
// 1x1 projection of a pooled branch (pooled_projection: conv -> pool -> shift -> ReLU); its buffer carries a
// zero halo of >= 1 (build() asks for it), so the taps are unconditional loads and only the divisor depends on
// the position.  One thread produces TWO horizontally adjacent outputs from a 3x4 window (12 loads instead of
// 18): the three column sums in the middle are shared.  Same arithmetic as conv_epilogue_avg (avg_finish).
__global__ void avgpool3s1_kernel(PoolArgs p) {
  const int cg = p.C / 8;
  const int H = p.ig.h, W = p.ig.w;
  const int W2 = (W + 1) >> 1;
  const size_t total = static_cast<size_t>(p.N) * cg * H * W2;
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i >= total) return;
  const int ow = 2 * static_cast<int>(i % W2);
  size_t t = i / W2;
  const int oh = t % H;
  t /= H;
  const int g = t % cg;
  const int n = t / cg;
  const bool two = ow + 1 < W;
  const float4* src = reinterpret_cast<const float4*>(p.in32) +
                      (((static_cast<size_t>(n) * p.ig.groups + g) * p.ig.hp + oh + p.ig.halo - 1) *
                           p.ig.wp + ow + p.ig.halo - 1) * 2;
  const int last = two ? 3 : 2;  // never read past the row's halo
  float col[4][8];
#pragma unroll
  for (int dw = 0; dw < 4; ++dw) {
    const int c = dw < 3 ? dw : last;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const float4 a = src[2 * c + hf], b = src[2 * (p.ig.wp + c) + hf], d = src[2 * (2 * p.ig.wp + c) + hf];
      col[dw][4 * hf + 0] = a.x + b.x + d.x;
      col[dw][4 * hf + 1] = a.y + b.y + d.y;
      col[dw][4 * hf + 2] = a.z + b.z + d.z;
      col[dw][4 * hf + 3] = a.w + b.w + d.w;
    }
  }
  const int rows = (oh > 0) + (oh < H - 1) + 1;
  float sh[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (p.shift != nullptr) {
    const float4 s0 = *reinterpret_cast<const float4*>(p.shift + g * 8);
    const float4 s1 = *reinterpret_cast<const float4*>(p.shift + g * 8 + 4);
    sh[0] = s0.x; sh[1] = s0.y; sh[2] = s0.z; sh[3] = s0.w;
    sh[4] = s1.x; sh[5] = s1.y; sh[6] = s1.z; sh[7] = s1.w;
  }
  const size_t at = ((static_cast<size_t>(n) * p.og.groups + p.out_goff + g) * p.og.hp + oh + p.og.halo) * p.og.wp +
                    ow + p.og.halo;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (k == 1 && !two) break;
    const int x = ow + k;
    const float inv = 1.0f / static_cast<float>(rows * ((x > 0) + (x < W - 1) + 1));
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o[j] = avg_finish(col[k][j], col[k + 1][j], col[k + 2][j], inv, sh[j], p.shift != nullptr);
    }
    if (p.out32 != nullptr) {
      float4* d = reinterpret_cast<float4*>(p.out32 + (at + k) * 8);
      d[0] = make_float4(o[0], o[1], o[2], o[3]);
      d[1] = make_float4(o[4], o[5], o[6], o[7]);
    } else {
      half8_t h;
#pragma unroll
      for (int j = 0; j < 8; ++j) h[j] = static_cast<_Float16>(o[j]);
      reinterpret_cast<half8_t*>(p.out)[at + k] = h;
      if (p.lo_out_groups > 0) {
        half8_t l;
#pragma unroll
        for (int j = 0; j < 8; ++j) l[j] = static_cast<_Float16>(o[j] - static_cast<float>(h[j]));
        reinterpret_cast<half8_t*>(p.out)[at + k + static_cast<size_t>(p.lo_out_groups) * p.og.hp * p.og.wp] = l;
      }
    }
  }
}

// GlobalAveragePooling2D + Dense(num_classes) + softmax, fp32.  Round 6: the last block's outputs arrive in
// float32 (BufferDesc::f32) -- the values the convolutions' accumulators held, not an fp16 copy of them.
__global__ __launch_bounds__(256) void head_kernel(const float* in, const float* w,
                                                   const float* b, const ExtPtrs* ext, size_t probs_off,
                                                   TensorGeom g, int K) {
  float* probs = ext->probs + probs_off;
  __shared__ float red[8][4];
  const int n = blockIdx.x;
  const int tid = threadIdx.x;
  const int C = g.groups * 8;
  const size_t plane = static_cast<size_t>(g.hp) * g.wp * 8;
  float part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const float* x = in + static_cast<size_t>(n) * g.groups * plane;
  const float invP = 1.0f / static_cast<float>(g.h * g.w);
  // one thread per 8-channel group: the map's pixels come in as whole 16-byte pieces
  for (int grp = tid; grp * 8 < C; grp += 256) {
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float4* xg = reinterpret_cast<const float4*>(x + static_cast<size_t>(grp) * plane);
    for (int y = 0; y < g.h; ++y)
      for (int xx = 0; xx < g.w; ++xx) {
        const float4 lo = xg[((y + g.halo) * g.wp + xx + g.halo) * 2], up = xg[((y + g.halo) * g.wp + xx + g.halo) * 2 + 1];
        s[0] += lo.x; s[1] += lo.y; s[2] += lo.z; s[3] += lo.w;
        s[4] += up.x; s[5] += up.y; s[6] += up.z; s[7] += up.w;
      }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float m = s[j] * invP;
      for (int k = 0; k < K; ++k) part[k] += m * w[static_cast<size_t>(grp * 8 + j) * K + k];
    }
  }
  for (int k = 0; k < K; ++k) {
    float v = part[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((tid & 63) == 0) red[k][tid >> 6] = v;
  }
  __syncthreads();
  if (tid == 0) {
    float logit[8], mx = -1e30f;
    for (int k = 0; k < K; ++k) {
      logit[k] = red[k][0] + red[k][1] + red[k][2] + red[k][3] + b[k];
      mx = fmaxf(mx, logit[k]);
    }
    float sum = 0.f;
    for (int k = 0; k < K; ++k) {
      logit[k] = expf(logit[k] - mx);
      sum += logit[k];
    }
    for (int k = 0; k < K; ++k) probs[static_cast<size_t>(n) * K + k] = logit[k] / sum;
  }
}

// ------------------------------------------------------------------ the graph

struct TensorRef {
  int buf = -1;  // index into buffers
  int h = 0, w = 0, c = 0;
};

struct BufferDesc {
  int h, w, c;  // channels = full (concat) width
  int halo = 0; // max padding any consumer needs (zero border kept in HBM)
  int min_examples = 1;  // imgconv tiles read whole groups of images: allocate at least this many
  bool f32 = false;      // float32 elements (a piece = 8 floats): tensors that no MFMA reads -- the raw 1x1 outputs of
                         // the pooled projections (input of an average pool) and the last block's outputs (input of the
                         // global pool) -- keep the accumulators' values instead of an fp16 rounding of them
  bool wide = false;     // precise mode (dv_model::precise): the tensor holds c / 8 groups of hi = fp16(x), then c / 8 groups
                         // of lo = fp16(x - hi); its consumers run their K over both with the same weights
  TensorGeom geom() const {
    return TensorGeom{h, w, halo, h + 2 * halo, w + 2 * halo, (wide ? 2 : 1) * (c / 8)};
  }
  size_t bytes_per_example() const {
    return static_cast<size_t>(h + 2 * halo) * (w + 2 * halo) * c * (f32 ? 4 : wide ? 4 : 2);
  }
};

enum OpType { kOpConv, kOpMaxPool, kOpAvgPool };

struct Op {
  OpType type;
  int in_buf, out_buf;
  int out_coff = 0;
  // conv
  int layer = -1;
  int kh = 0, kw = 0, stride = 1, pad_h = 0, pad_w = 0;
  int cin = 0, cin_real = 0, cout = 0;
  bool in_wide = false;          // the input tensor holds hi + lo pieces (BufferDesc::wide): every K chunk is multiplied
                                 // against both (ConvArgs::wide_in, conv_slab_wide)
  int ih = 0, iw = 0, oh = 0, ow = 0;
  int nb = 4;
  int n_steps = 0, n_chunks = 0;
  size_t w_off = 0;      // halfs into packed weights
  size_t shift_off = 0;  // floats into shifts
  size_t tbl_off = 0;    // int2 entries into the chunk tables
  bool raw = false;              // conv: skip shift + ReLU (applied by a later pool)
  int group_followers = 0;       // conv: the next k ops are siblings sharing this launch
  bool first_u8 = false;         // conv: reads the uint8 image directly (fused preprocess)
  bool pool_shift_relu = false;  // avgpool: add shift[c] and ReLU after averaging
  bool pool_in = false;          // 1x1 conv that max-pools (3x3, stride 2) its input on the fly
  int side_pool_partner = -1;    // 3x3 / 2 conv <-> the sibling max-pool it computes on the side (choose_side_pool)
  int avg_partner = -1;          // raw 1x1 conv <-> the average pool behind it, taken in the launch's epilogue (choose_avg_epilogue)
  int avg_tile_g = 0;            // leader of such a launch: whole maps per 256-pixel block
  bool pool_out = false;         // conv whose output is max-pooled (3x3, stride 2) before it is stored
                                 // (conv_pool_resident_kernel; oh / ow stay the conv's, the buffer is pooled)
  // Fused stem (stem.hip): the op marked stem_a / stem_b runs together with the op that
  // follows it as ONE launch; the tensor between them is never materialised.
  // imgconv.hip: whole-map tiles, both operands through LDS (set on the launch's leader op)
  int band = 0;                  // conv_mfma_kernel's row-band mode: map rows (= taps kept), 0 = off
  bool split = false;            // the LAUNCH carries W_hi + W_lo weight images (choose_split) ...
  bool split_rows = false;       // ... and this op's couts are among them (siblings of a group may not be)
  int split_tiles = 0;           // leader: leading cout tiles of the launch that hold (hi, lo) pairs
  bool v2 = false;
  int v2_g = 0;                  // images per tile
  int v2_steps = 0;              // K steps (KC channel chunks each)
  int v2_tiles = 0;              // cout tiles of nb*32
  bool stem_a = false;           // first conv (uint8 input) + conv 3x3 32->32
  bool stem_b = false;           // conv 3x3 32->64 + maxpool 3x3/2 + conv 1x1 64->80
  // chain.hip: this op and the chain_len - 1 ops behind it (1 x k / k x 1, each reading its
  // predecessor) run as ONE launch; the tensors between them live in LDS only
  int chain_len = 0;
  int chain_g = 0;               // images per tile
  int chain_tpx = 0;             // pixels per tile (192: small maps, 1-D filters; 256: 35x35 stage, 3x3 / 5x5)
  bool in_chain = false;         // a non-leading member of a chain
};

struct LayerInfo {
  int kh, kw, cin, cout;
  int64_t param_off;
};

}  // namespace

struct dv_model {
  int device = 0;
  dv_model_desc desc{};
  std::vector<BufferDesc> buffers;
  std::vector<Op> ops;
  std::vector<LayerInfo> layers;  // convs then dense
  int64_t n_params = 0;
  int feat_buf = -1, feat_p = 0, feat_c = 0;
  int stem_ops_end = 0, stem_out_buf = -1;
  int stem_a_grid = 512, stem_b_grid = 256;  // persistent grids of the fused stem kernels
  int n_cus = 256;
  size_t packed_halfs = 0, shift_floats = 0, tbl_entries = 0;
  std::vector<dv::DeviceBuffer> dbuf;
  dv::DeviceBuffer d_w, d_shift, d_dense_w, d_dense_b, d_tbl;
  // Blank-row skipping through the stem (round 6: on by default, DV_BLANK_SKIP=0 / dv_model_set_blank_skip turn it
  // off): tiles of conv2 / stem_b / the 3x3 80->192 whose receptive field holds only the zero rows below
  // the pile-up are copied from the all-blank image's response instead of computed -- bit-identical.
  bool blank_skip = false;        // applicable to this model and not switched off by the environment
  bool blank_enabled = true;      // dv_model_set_blank_skip
  bool blank_ready = false;       // the blank responses have been computed (after load_weights)
  int blank_conv4_op = -1;        // op index of the stem's 3x3 80->192
  dv::DeviceBuffer d_blank_thr;   // int32 [7][max_batch], blank_rows_kernel + blank_need_kernel
  dv::DeviceBuffer d_blank_conv4; // the 3x3 80->192's output (pooled when its kernel pools) for the all-blank image (one example)
  dv::DeviceBuffer d_blank_c2;    // conv2's output for the all-blank image
  dv::DeviceBuffer d_blank_b;     // stem_b's (the 1x1 64->80's) output for the all-blank image
  bool blank_on() const { return blank_ready && blank_enabled; }
  // Precise mode (round 6): every fp16 tensor of the 17x17 and 8x8 stages is stored as hi + lo fp16 pieces
  // and its consumers multiply both (K doubled, the factorised-7x7 chains run per layer) -- what it takes to hold 1e-3
  // on every long-read seed, at about +40 % of the forward.  Default: on for > 8 input channels (dv_model_create).
  bool precise = false;
  bool wide_stage = false;        // build(): buffers created now belong to the wide stages
  bool loaded = false;
  std::vector<float> h_shift, h_dense_b;   // as computed by dv_model_load_weights (before any calibration)
  dv::DeviceBuffer d_ext;         // ExtPtrs: the caller's image / probability pointers of the running forward
  struct GraphEntry {
    int n;
    hipStream_t stream;
    hipGraphExec_t exec;
  };
  std::vector<GraphEntry> graphs;  // captured forwards, see dv_model_infer
  int64_t graph_captures = 0, graph_replays = 0;

  // ---- builder ------------------------------------------------------------
  int new_buffer(int h, int w, int c) {
    buffers.push_back({h, w, c, 0});
    buffers.back().wide = precise && wide_stage;
    return static_cast<int>(buffers.size()) - 1;
  }
  static int pick_nb(int cout) {
    // DV_NB6: 192-cout single tiles for 129..192-cout layers (see launch_conv6)
    static const bool nb6 = getenv("DV_NB6") != nullptr && atoi(getenv("DV_NB6")) != 0;
    if (nb6 && cout > 128 && cout <= 192) return 6;
    // Cost of a cout tiling ~ tiles x (nb MFMA columns + 1 pixel-fragment stream): a
    // 160-wide layer is cheaper as 2 x 96 (one sixth padding) than as 5 x 32, whose
    // blocks re-load every pixel fragment five times.  Ties -> less padding.
    int best = 4, best_cost = 1 << 30, best_waste = 1 << 30;
    for (int nb = 4; nb >= 1; --nb) {
      const int bn = nb * 32;
      const int tiles = (cout + bn - 1) / bn;
      const int cost = tiles * (nb + 1), waste = tiles * bn - cout;
      if (cost < best_cost || (cost == best_cost && waste < best_waste)) {
        best_cost = cost;
        best_waste = waste;
        best = nb;
      }
    }
    return best;
  }
  TensorRef conv(TensorRef x, int cout, int kh, int kw, int stride = 1, bool same = true,
                 int dst_buf = -1, int dst_coff = 0, int cin_real = -1) {
    Op op;
    op.type = kOpConv;
    op.kh = kh;
    op.kw = kw;
    op.stride = stride;
    op.pad_h = same ? (kh - 1) / 2 : 0;
    op.pad_w = same ? (kw - 1) / 2 : 0;
    op.in_wide = buffers[x.buf].wide;
    op.cin = x.c;
    op.cin_real = cin_real < 0 ? x.c : cin_real;
    op.cout = cout;
    op.ih = x.h;
    op.iw = x.w;
    op.oh = (x.h + 2 * op.pad_h - kh) / stride + 1;
    op.ow = (x.w + 2 * op.pad_w - kw) / stride + 1;
    op.in_buf = x.buf;
    if (dst_buf < 0) {
      dst_buf = new_buffer(op.oh, op.ow, cout);
      dst_coff = 0;
    }
    op.out_buf = dst_buf;
    op.out_coff = dst_coff;
    op.nb = pick_nb(cout);
    op.n_chunks = kh * kw * (op.cin / kChunk);
    op.n_steps = (op.n_chunks + kSlabChunks - 1) / kSlabChunks;  // weight slabs
    op.shift_off = shift_floats;
    shift_floats += cout + 128;  // padded: the epilogue reads whole 32-cout tiles
    op.tbl_off = tbl_entries;
    tbl_entries += op.n_chunks;
    op.layer = static_cast<int>(layers.size());
    layers.push_back({kh, kw, op.cin_real, cout, n_params});
    n_params += static_cast<int64_t>(kh) * kw * op.cin_real * cout + 3LL * cout;
    ops.push_back(op);
    TensorRef out;
    out.buf = dst_buf;
    out.h = op.oh;
    out.w = op.ow;
    out.c = cout;  // view width; the consumer of a concat reads the full buffer
    return out;
  }
  TensorRef full(int buf) const {
    TensorRef t;
    t.buf = buf;
    t.h = buffers[buf].h;
    t.w = buffers[buf].w;
    t.c = buffers[buf].c;
    return t;
  }
  // AveragePooling2D(3,1,'same') -> conv 1x1 -> BN -> ReLU, evaluated as
  // conv 1x1 (raw) -> average pool -> +shift -> ReLU.  A 1x1 convolution is a
  // per-pixel linear map, so it commutes with the (per-pixel-normalised)
  // average; pooling the Cout (32..192) projected channels instead of the Cin
  // (192..2048) input channels moves 4-10x fewer bytes.
  void pooled_projection(TensorRef x, int cout, int dst_buf, int dst_coff) {
    TensorRef raw = conv(x, cout, 1, 1);
    ops.back().raw = true;                   // no shift, no ReLU in the conv epilogue
    buffers[raw.buf].f32 = true;             // averaged in float32 (conv_epilogue_avg / avgpool3s1_kernel)
    buffers[raw.buf].wide = false;
    const size_t shift_off = ops.back().shift_off;
    pool(kOpAvgPool, raw, dst_buf, dst_coff);
    ops.back().shift_off = shift_off;        // applied after the pool
    ops.back().pool_shift_relu = true;
  }
  TensorRef pool(OpType type, TensorRef x, int dst_buf = -1, int dst_coff = 0) {
    Op op;
    op.type = type;
    op.in_buf = x.buf;
    op.ih = x.h;
    op.iw = x.w;
    op.cin = x.c;
    op.cout = x.c;
    if (type == kOpMaxPool) {
      op.oh = (x.h - 3) / 2 + 1;
      op.ow = (x.w - 3) / 2 + 1;
    } else {
      op.oh = x.h;
      op.ow = x.w;
    }
    if (dst_buf < 0) {
      dst_buf = new_buffer(op.oh, op.ow, x.c);
      dst_coff = 0;
    }
    op.out_buf = dst_buf;
    op.out_coff = dst_coff;
    ops.push_back(op);
    TensorRef out;
    out.buf = dst_buf;
    out.h = op.oh;
    out.w = op.ow;
    out.c = x.c;
    return out;
  }

  // Sibling 1x1 convolutions of an Inception block read the same tensor.  Hoist
  // them next to the first one and mark them as ONE launch (ConvArgs::br): the
  // input is then fetched from HBM once and re-read from L2 by the siblings.
  // Layer (= weight) order is untouched -- only the execution order changes,
  // which is legal because every hoisted op depends on the shared input only.
  void group_siblings() {
    static const bool off = getenv("DV_NO_GROUPING") != nullptr;  // tuning knob
    if (off) return;
    for (size_t i = 0; i < ops.size(); ++i) {
      Op& lead = ops[i];
      if (lead.type != kOpConv || lead.kh != 1 || lead.kw != 1 || lead.stride != 1) continue;
      std::vector<size_t> sib;
      for (size_t j = i + 1; j < ops.size() && j < i + 16 && sib.size() + 1 < kMaxBranches; ++j) {
        const Op& o = ops[j];
        if (o.type == kOpConv && o.kh == 1 && o.kw == 1 && o.stride == 1 &&
            o.in_buf == lead.in_buf) {
          sib.push_back(j);
        }
      }
      if (sib.empty()) continue;
      // tile width over the concatenated cout space (32-cout subtiles), same cost model
      // as pick_nb: tiles x (nb MFMA columns + 1 pixel-fragment stream)
      int subs = (lead.cout + 31) / 32;
      for (size_t j : sib) subs += (ops[j].cout + 31) / 32;
      // 128-cout tiles (<4,2>, two blocks per CU since round 2) for every grouped head from 7
      // subtiles up: the 35x35 heads (7-8 subtiles) then take 2 tiles instead of 3 -- the input is
      // re-read twice instead of three times and no padding subtile is multiplied: 495 / 556 / 583
      // -> 448 / 522 / 567 us, +0.8 % end to end (round 4, tools/r4_run.sh ab:DV_HEADS_NB4_MIN=7;
      // round 2 had measured 96-cout tiles faster when <4,2> still ran one block per CU).
      // heads that max-pool their input on the fly (DV_NO_POOL2_IN_CONV): ONE tile of all 7 subtiles, so
      // that every 3x3 window is fetched and reduced once
      static const int nb4_min = getenv("DV_HEADS_NB4_MIN") ? atoi(getenv("DV_HEADS_NB4_MIN")) : 7;  // tuning knob
      const int nb = lead.pool_in && subs == 7 && getenv("DV_POOL2_NB3") == nullptr ? 7 : subs >= nb4_min ? 4 : 3;
      std::vector<Op> moved;
      for (size_t j : sib) moved.push_back(ops[j]);
      for (size_t k = sib.size(); k-- > 0;) ops.erase(ops.begin() + sib[k]);
      ops.insert(ops.begin() + i + 1, moved.begin(), moved.end());
      ops[i].group_followers = static_cast<int>(moved.size());
      for (size_t k = 0; k <= moved.size(); ++k) ops[i + k].nb = nb;
      i += moved.size();
    }
  }

  // Tile geometry of an imgconv launch for `g` images per tile.
  dv::ImgConvArgs imgconv_geometry(const Op& op, int g) const {
    dv::ImgConvArgs a{};
    const int kc = dv::imgconv_kc(op.kh, op.kw);
    a.G = g;
    a.P = op.oh * op.ow;
    a.RP = op.oh + op.kh - 1;
    a.CP = op.ow + op.kw - 1;
    a.plane_pieces = g * a.RP * a.CP;
    a.act_pieces = kc * 2 * a.plane_pieces;
    a.act_slab_bytes = (a.act_pieces * 16 + 1023) / 1024 * 1024;
    a.n_steps = (op.cin / kChunk + kc - 1) / kc;
    a.c.KH = op.kh;
    a.c.KW = op.kw;
    return a;
  }
  // Stride-1 convolutions on small maps run in imgconv.hip (whole-map tiles, both operands
  // in LDS) when a tile of G images fills at least 3/4 of the 512-pixel tile and the double
  // buffered slabs fit the CU's LDS.  DV_NO_IMGCONV keeps conv_mfma_kernel for all of them.
  void choose_imgconv() {
    if (getenv("DV_NO_IMGCONV") != nullptr) return;
    const char* only = getenv("DV_IMGCONV_TAPS");  // tuning knob: e.g. "9,25" = only 3x3 and 5x5
    for (size_t i = 0; i < ops.size(); ++i) {
      Op& op = ops[i];
      const int followers = op.type == kOpConv ? op.group_followers : 0;
      bool f32_out = false;   // float32 outputs go through conv_epilogue only
      for (int gi = 0; gi <= followers; ++gi) f32_out = f32_out || buffers[ops[i + gi].out_buf].f32;
      if (op.type == kOpConv && !f32_out && !op.in_wide && !buffers[op.out_buf].wide && !op.first_u8 && !op.pool_in && !op.pool_out && !op.stem_a && !op.stem_b &&
          op.chain_len == 0 && !op.in_chain &&
          !(i > 0 && (ops[i - 1].stem_a || ops[i - 1].stem_b)) && op.stride == 1 &&
          dv::imgconv_supported(op.kh, op.kw, op.nb) && op.oh * op.ow <= 512) {
        // 1x1 layers have no taps to share a patch between: the DMA count equals
        // conv_mfma_kernel's fragment loads and the LDS round trip only costs (measured
        // 0.75x); they stay on conv_mfma_kernel unless DV_IMGCONV_1X1 is set.
        bool wanted = op.kh * op.kw > 1 || getenv("DV_IMGCONV_1X1") != nullptr;
        // Measured at 8 K examples: +10..30 % on the 10x25 maps (3x3, 5x5), no gain on the
        // 4x12 maps (7-tap filters) and a loss on 1x5 (the patch is mostly halo): only maps
        // of at least DV_IMGCONV_MINP pixels (default 100) take this path.
        static const int min_p = getenv("DV_IMGCONV_MINP") ? atoi(getenv("DV_IMGCONV_MINP")) : 100;
        if (op.oh * op.ow < min_p) wanted = false;
        if (only != nullptr) {
          wanted = false;
          for (const char* q = only; *q;) {
            if (atoi(q) == op.kh * op.kw) wanted = true;
            while (*q && *q != ',') ++q;
            if (*q == ',') ++q;
          }
        }
        int subs = 0;
        for (int gi = 0; gi <= followers; ++gi) subs += (ops[i + gi].cout + 31) / 32;
        const int P = op.oh * op.ow;
        for (int g = 512 / P; wanted && g >= 1 && g * P >= 384; --g) {
          const dv::ImgConvArgs a = imgconv_geometry(op, g);
          if (a.act_slab_bytes > 64 * 1024 || dv::imgconv_lds_bytes(a, op.nb) > 160 * 1024) continue;
          op.v2 = true;
          op.v2_g = g;
          op.v2_steps = a.n_steps;
          op.v2_tiles = (subs + op.nb - 1) / op.nb;
          buffers[op.in_buf].min_examples = std::max(buffers[op.in_buf].min_examples, g);
          break;
        }
      }
      i += followers;
    }
  }

  // Filters taller than the map: conv_mfma_kernel's row-band mode (ConvArgs::band) skips the
  // taps that only ever see the zero halo.  At 100 x 221 inputs these are the 7x1 layers of
  // the 4 x 12 maps (4 of 7 taps remain) and the 3x3 / 3x1 layers of the 1 x 5 maps (the
  // middle row only).  Needs every output row to see ALL map rows (so each row keeps exactly
  // H taps): H <= min(pad, KH - 1 - pad) + 1.  DV_NO_BAND keeps the full filters.
  void choose_band() {
    if (getenv("DV_NO_BAND") != nullptr) return;
    for (Op& op : ops) {
      if (op.type != kOpConv || op.first_u8 || op.pool_in || op.pool_out || op.stem_a || op.stem_b || op.v2 ||
          op.chain_len != 0 || op.in_chain || op.group_followers != 0 || op.stride != 1 || op.kh <= 1 || op.oh != op.ih) {
        continue;
      }
      const int h = op.ih;
      if (h >= op.kh || h > std::min(op.pad_h, op.kh - 1 - op.pad_h) + 1 || op.ow < 5) continue;
      bool follower = false;  // a sibling inside another op's launch keeps that launch's geometry
      for (const Op& lead : ops) {
        if (lead.type == kOpConv && lead.group_followers > 0 && &op > &lead &&
            &op <= &lead + lead.group_followers) {
          follower = true;
        }
      }
      if (follower) continue;
      op.band = h;
      // the row-band 7x1 layers are the one shape where a single 192-cout tile (launch_conv6)
      // measured faster than two 96-cout tiles (-5...-13 %); DV_NO_BAND_NB6 keeps two tiles
      if (op.cout > 128 && op.cout <= 192 && op.nb == 3 && getenv("DV_NO_BAND_NB6") == nullptr) op.nb = 6;
      op.n_chunks = h * op.kw * (op.cin / kChunk);
      op.n_steps = (op.n_chunks + kSlabChunks - 1) / kSlabChunks;
    }
  }

  // The reduction block mixed3 runs MaxPooling2D(3, 2) next to a 3x3 / stride-2 'valid' convolution of the
  // SAME tensor: per 16-channel chunk the convolution's nine tap fragments are exactly the pool's
  // window pieces, so the workgroups of its cout tile 0 take the maximum on the side (SidePool) and the
  // pool's own launch (0.27 ms, a full re-read of the block input) disappears.  mixed8's pool has no
  // such sibling (its stride-2 convolutions read the 1x1 outputs).  DV_NO_SIDE_POOL keeps the launch.
  void choose_side_pool() {
    if (getenv("DV_NO_SIDE_POOL") != nullptr) return;
    for (size_t pi = 0; pi < ops.size(); ++pi) {
      Op& pl = ops[pi];
      if (pl.type != kOpMaxPool) continue;
      for (size_t ci = 0; ci < ops.size(); ++ci) {
        Op& cv = ops[ci];
        if (cv.type != kOpConv || cv.in_buf != pl.in_buf || cv.stride != 2 || cv.kh != 3 || cv.kw != 3 ||
            cv.pad_h != 0 || cv.pad_w != 0 || cv.nb != 4 || cv.group_followers != 0 || cv.first_u8 || cv.pool_in ||
            cv.pool_out || cv.stem_a || cv.stem_b || cv.v2 || cv.band || cv.split || cv.chain_len != 0 || cv.in_chain ||
            cv.raw || cv.cin != pl.cin || cv.cin % kChunk != 0 || cv.cin != buffers[cv.in_buf].c ||
            cv.oh != pl.oh || cv.ow != pl.ow || cv.side_pool_partner >= 0) {
          continue;
        }
        cv.side_pool_partner = static_cast<int>(pi);
        pl.side_pool_partner = static_cast<int>(ci);
        break;
      }
    }
  }

  // Split weights (HISTORY.md 15).  The fp16 rounding of the BN-folded weights is ~3/4 of the variance
  // of the CNN's error against the fp32 reference (tools/r4_layer_sensitivity.py: a flat budget, no
  // layer above 3.5 %), and it is the half that a kernel can remove without touching its pixel
  // operand.  Selected conv_mfma_kernel launches therefore carry W as W_hi + W_lo (both fp16,
  // W_lo = fp16(W - W_hi)): the packed image holds every K chunk twice and the kernel multiplies the
  // same pixel fragment by both -- products are exact, the sum is fp32, so those layers compute
  // with 22-bit weights.  Which: in the 17x17 blocks the two 1x1 layers whose output is block
  // output (b1, pooled projection -- the leading cout tiles of the grouped heads launch), in
  // mixed8..10 every 1x1 / 3-tap / 3x3 layer.  Measured on 2048 pileups x seeds 17 / 29
  // (profiles/r04_precision_sweep.txt): max |dp| 1.15e-3 / 1.55e-3 without, 7.8e-4 / 8.6e-4 with.
  // DV_SPLIT_FROM=<layer> (construction order; 94 = none, 0 = every conv_mfma layer) and
  // DV_SPLIT_LAYERS=<list> override the set for A/B runs.
  void choose_split() {
    const int first_layer_env = getenv("DV_SPLIT_FROM") ? atoi(getenv("DV_SPLIT_FROM")) : -1;  // per model (tests)
    // mixed4 (the 17x17 stage) starts at conv layer 30 of the 94 (5 stem + 3 x 7 + 4), mixed8 at 70.
    // The default set is a property of the LAYER, not of the kernel that happens to run it: 1x1
    // layers from mixed4 on, 3-tap and 3x3 layers from mixed8 on -- never the factorised-7x7
    // layers, which run as fused chains (and must give the same bits when DV_NO_CHAIN unfuses them).
    const char* list_env = getenv("DV_SPLIT_LAYERS");   // experiments: an explicit comma list of layers
    const char* split_default_env = getenv("DV_SPLIT_DEFAULT");   // 1 = the round-4 default set below
    auto wanted = [&](const Op& o) {
      if (list_env != nullptr) {
        for (const char* q = list_env; *q;) {
          if (atoi(q) == o.layer) return true;
          while (*q && *q != ',') ++q;
          if (*q == ',') ++q;
        }
        return false;
      }
      if (first_layer_env >= 0) return o.layer >= first_layer_env;
      // Round 5: OFF unless DV_SPLIT_DEFAULT=1.  The shift calibration (dv_model_calibrate, calib.hip) removes
      // the per-channel mean of the weight AND activation rounding at no run-time cost and measures better on
      // every held-out seed at N = 65,536 than this set did (profiles/r05_cnn_tail.txt: max |dp| 8.6e-4 /
      // 2.9e-4 / 7.2e-4 calibrated without split weights against 1.01e-3 / 4.2e-4 / 9.1e-4 with them).
      if (split_default_env == nullptr || atoi(split_default_env) == 0) return false;
      // 17x17 stage: the two 1x1 layers of a block whose output IS block output -- the b1 branch
      // (written into the concat buffer) and the pooled projection (raw) -- not the heads of the
      // factorised-7x7 branches (measured: profiles/r04_precision_sweep.txt)
      if (o.kh * o.kw == 1 && o.layer >= 30 && o.layer < 70) return o.raw || buffers[o.out_buf].c > o.cout;
      return o.layer >= 70 && std::max(o.kh, o.kw) <= 3;
    };
    for (size_t i = 0; i < ops.size(); ++i) {
      Op& op = ops[i];
      if (op.type != kOpConv) continue;
      const int followers = op.group_followers;
      const bool eligible = !op.first_u8 && !op.pool_in && !op.pool_out && !op.stem_a && !op.stem_b && !op.v2 &&
                            op.chain_len == 0 && !op.in_chain && !(i > 0 && (ops[i - 1].stem_a || ops[i - 1].stem_b)) &&
                            op.nb <= 4 && static_cast<int>(i) != blank_conv4_op && !op.in_wide;
      int n_wanted = 0;
      for (int gi = 0; gi <= followers; ++gi) n_wanted += wanted(ops[i + gi]) ? 1 : 0;
      if (eligible && n_wanted > 0) {
        // Siblings of which only some are wanted: the wanted ones go to the front of the launch's
        // cout space, and if they fill whole cout tiles only those tiles carry (hi, lo) pairs
        // (ConvArgs::split_tiles); otherwise -- or when the leader itself is not wanted -- the whole
        // launch is split.
        int split_subs = 0, all_subs = 0;
        bool partial = n_wanted <= followers && wanted(op);
        if (partial) {
          std::stable_partition(ops.begin() + i + 1, ops.begin() + i + 1 + followers,
                                [&](const Op& o) { return wanted(o); });
          for (int gi = 0; gi <= followers; ++gi) {
            if (wanted(ops[i + gi])) split_subs += (ops[i + gi].cout + 31) / 32;
          }
          partial = split_subs % ops[i].nb == 0;
        }
        for (int gi = 0; gi <= followers; ++gi) all_subs += (ops[i + gi].cout + 31) / 32;
        Op& lead = ops[i];   // (stable_partition leaves the leader in place)
        lead.split_tiles = partial ? split_subs / lead.nb : (all_subs + lead.nb - 1) / lead.nb;
        for (int gi = 0; gi <= followers; ++gi) {
          Op& o = ops[i + gi];
          o.split = true;
          o.split_rows = !partial || wanted(o);
          o.n_chunks *= 2;
          o.n_steps = (o.n_chunks + kSlabChunks - 1) / kSlabChunks;
        }
      }
      i += followers;
    }
  }

  // Pooled projections (conv 1x1 raw -> AveragePooling2D(3, 1, 'same') -> shift -> ReLU): when the heads
  // launch that holds the raw 1x1 runs 128-cout tiles and whole maps fill a 256-pixel block to >= 90 %
  // (10x25 = 250 pixels: 1 map; 4x12: 5 maps; 1x5: 51 maps), its blocks are laid over whole maps
  // (ConvArgs::tile_g) and the pool happens in the epilogue (conv_epilogue_avg): the raw tensor is never
  // written and the avg-pool launch is gone.  Not for split launches (their own kernel variants), not for
  // heads that pool their input on the fly.  DV_NO_AVG_EPI keeps conv -> avgpool3s1_kernel (same bits).
  void choose_avg_epilogue() {
    if (getenv("DV_NO_AVG_EPI") != nullptr) return;
    const int min_g = getenv("DV_AVG_EPI_MIN_G") ? atoi(getenv("DV_AVG_EPI_MIN_G")) : 1;   // tuning knob: whole maps per block
    // whole maps must fill this share of the 256 pixel slots (percent).  Round 5: 90 (the ILLUMINA30 maps: 98 / 94 / 100 %);
    // round 6: 85, which takes in ONT_R104's 10x22 maps (86 %) -- PACBIO's 10x16 (62 %) keeps the separate pool
    const int min_fill = getenv("DV_AVG_EPI_MIN_FILL") ? atoi(getenv("DV_AVG_EPI_MIN_FILL")) : 85;
    for (size_t i = 0; i < ops.size(); ++i) {
      Op& lead = ops[i];
      if (lead.type != kOpConv) continue;
      const int followers = lead.group_followers;
      const int px = lead.oh * lead.ow;
      const bool ok = lead.kh == 1 && lead.kw == 1 && lead.stride == 1 && lead.nb == 4 && !lead.split && !lead.pool_in &&
                      !lead.pool_out && !lead.v2 && !lead.band && lead.chain_len == 0 && !lead.in_chain &&
                      !lead.first_u8 && !lead.stem_a && !lead.stem_b && px >= 5 && px <= 256 &&
                      (256 / px) * px * 100 >= 256 * min_fill && 256 / px >= min_g;
      if (ok) {
        for (int gi = 0; gi <= followers; ++gi) {
          Op& c = ops[i + gi];
          if (!c.raw) continue;
          for (size_t j = i + followers + 1; j < ops.size(); ++j) {
            Op& pl = ops[j];
            if (pl.type == kOpAvgPool && pl.in_buf == c.out_buf && pl.pool_shift_relu && pl.avg_partner < 0) {
              c.avg_partner = static_cast<int>(j);
              pl.avg_partner = static_cast<int>(i + gi);
              lead.avg_tile_g = 256 / px;
              break;
            }
          }
        }
      }
      i += followers;
    }
  }

  // Chains of stride-1 'same' convolutions in which every layer reads only its predecessor run in
  // chain.hip, intermediates in LDS:
  //   * maps of <= 96 pixels (the 17x17 stage at WGS width), 1 x k / k x 1 filters: the factorised
  //     7x7 branches of mixed4..mixed8 -- G whole maps in a 192-pixel tile, two layers or more;
  //   * maps of 97..256 pixels (the 35x35 stage), 3x3 / 5x5 filters with 64..96 couts: the
  //     3x3 -> 3x3 branch of mixed0..2, and the single 5x5 / 3x3 layers next to it (one-layer
  //     "chains": both operands from LDS, loader waves) -- one or two maps in a 256-pixel tile.
  // A tile must be at least two thirds full and the activation tile plus two weight slabs must
  // fit the CU's LDS.  DV_NO_CHAIN keeps the per-layer kernels; DV_NO_CHAIN2D only those of the
  // 35x35 stage; DV_CHAIN2D_MIN_LEN=1 also takes its single layers from imgconv.
  void choose_chains() {
    if (getenv("DV_NO_CHAIN") != nullptr) return;
    const bool no_2d = getenv("DV_NO_CHAIN2D") != nullptr;
    // measured (profiles/r03_chain2d_ab.txt): the 3x3 -> 3x3 pairs gain 6 % over two imgconv launches;
    // single layers lose 5-30 % to imgconv (its tiles of two maps pipeline the next tile's input, a
    // one-layer chain exposes it), so they stay there unless DV_CHAIN2D_MIN_LEN=1
    const int min_len_2d = getenv("DV_CHAIN2D_MIN_LEN") ? atoi(getenv("DV_CHAIN2D_MIN_LEN")) : 2;
    std::vector<int> readers(buffers.size(), 0);
    for (const Op& o : ops) readers[o.in_buf]++;
    auto plain = [&](const Op& o) {
      return o.type == kOpConv && !buffers[o.out_buf].f32 && !o.in_wide && !buffers[o.out_buf].wide && o.stride == 1 && (o.kh & 1) && (o.kw & 1) && o.kh * o.kw > 1 &&
             o.pad_h == (o.kh - 1) / 2 && o.pad_w == (o.kw - 1) / 2 && o.group_followers == 0 &&
             !o.first_u8 && !o.pool_in && !o.pool_out && !o.raw && !o.stem_a && !o.stem_b && o.cin % kChunk == 0 &&
             o.cin == o.cin_real && o.oh == o.ih && o.ow == o.iw;
    };
    auto one_d = [&](const Op& o) {
      return plain(o) && (o.kh == 1) != (o.kw == 1) && std::max(o.kh, o.kw) <= dv::kChainMaxTaps;
    };
    auto two_d = [&](const Op& o) {
      const int subs = (o.cout + 31) / 32;
      return plain(o) && ((o.kh == 3 && o.kw == 3) || (o.kh == 5 && o.kw == 5)) && subs >= 2 && subs <= 3;
    };
    for (size_t i = 0; i < ops.size(); ++i) {
      const int P = ops[i].oh * ops[i].ow;
      const bool big = P > dv::kChainTilePx / 2;
      if (big ? (no_2d || P > dv::kChainTilePxBig || !two_d(ops[i])) : !one_d(ops[i])) continue;
      auto member = [&](const Op& o) { return big ? two_d(o) : one_d(o); };
      size_t len = 1;
      while (i + len < ops.size() && len < static_cast<size_t>(dv::kChainMaxLayers)) {
        const Op& prev = ops[i + len - 1];
        const Op& next = ops[i + len];
        if (!member(next) || next.in_buf != prev.out_buf || prev.out_coff != 0 || readers[prev.out_buf] != 1 ||
            prev.cout % 32 != 0 || buffers[prev.out_buf].c != prev.cout) {
          break;
        }
        ++len;
      }
      if (static_cast<int>(len) < (big ? min_len_2d : 2)) continue;
      const int tpx = big ? dv::kChainTilePxBig : dv::kChainTilePx;
      const int g = tpx / P;
      if (g * P < tpx * 2 / 3) continue;
      size_t act = 0, slot = 0;
      bool fits = true;
      for (size_t k = 0; k < len; ++k) {
        const Op& o = ops[i + k];
        act = std::max(act, static_cast<size_t>(o.cin / 8) * tpx * 16);
        slot = std::max(slot, static_cast<size_t>(o.kh * o.kw) * 2 * ((o.cout + 31) / 32 * 32) * 16);
        // the 192-pixel shape halves the couts between two waves: 4..6 subtiles of 32
        if (!big) fits = fits && (o.cout + 31) / 32 >= 4 && (o.cout + 31) / 32 <= 6;
      }
      if (!fits || act + 2 * slot + 16 > 160 * 1024) continue;
      ops[i].chain_len = static_cast<int>(len);
      ops[i].chain_g = g;
      ops[i].chain_tpx = tpx;
      for (size_t k = 1; k < len; ++k) {
        ops[i + k].in_chain = true;
        const int c = ops[i + k - 1].cout;
        buffers[ops[i + k].in_buf] = {1, 1, c, 0};   // LDS only
      }
      buffers[ops[i].in_buf].min_examples = std::max(buffers[ops[i].in_buf].min_examples, g);
      i += len - 1;
    }
  }

  // tf_keras applications/inception_v3.py, construction order = layer order.
  void build() {
    const int in_buf = new_buffer(desc.height, desc.width, 16);
    TensorRef x = full(in_buf);
    x = conv(x, 32, 3, 3, 2, false, -1, 0, desc.channels);
    if (desc.channels <= 16 && getenv("DV_NO_U8_CONV1") == nullptr &&
        (desc.channels <= 8 || getenv("DV_NO_U8_CONV1_WIDE") == nullptr)) {
      Op& f = ops.back();
      f.first_u8 = true;  // conv_first_u8_kernel: K chunk = 2 taps x 8 channels (C <= 8), 1 tap x 16 (C <= 16)
      f.nb = 1;
      f.n_chunks = desc.channels <= 8 ? (f.kh * f.kw + 1) / 2 : f.kh * f.kw;
      f.n_steps = 1;
      buffers[in_buf] = {1, 1, 16, 0};  // the fp16 staging image is never materialised
    }
    x = conv(x, 32, 3, 3, 1, false);
    x = conv(x, 64, 3, 3);
    // (layer order: the two remaining stem convs are created before the pools run)
    if (getenv("DV_NO_POOL_FUSE") == nullptr) {  // tuning knob
      // max-pool fused into the 1x1 that consumes it (conv_pool1x1_kernel)
      TensorRef pooled = x;
      pooled.h = (x.h - 3) / 2 + 1;
      pooled.w = (x.w - 3) / 2 + 1;
      const int ih = x.h, iw = x.w;
      x = conv(pooled, 80, 1, 1, 1, false);
      ops.back().pool_in = true;
      ops.back().ih = ih;
      ops.back().iw = iw;
    } else {
      x = pool(kOpMaxPool, x);
      x = conv(x, 80, 1, 1, 1, false);
    }
    // Fused stem kernels (stem.hip): conv1+conv2 and conv3+maxpool+1x1 as two persistent
    // launches whose intermediates stay in LDS.  DV_NO_STEM_FUSE keeps the per-layer path
    // (also used for inputs with more than 8 channels).
    if (getenv("DV_NO_STEM_FUSE") == nullptr) {
      // stem_a reads the uint8 image with two taps x 8 channels per chunk (C <= 8) or, round 6, one tap x 16
      // channels (C = 9..12: the long-read channel sets ONT_R104 9, PACBIO 10; DV_NO_STEM_A_WIDE keeps
      // conv_first_u8 (wide) + a per-layer conv2 for them).  stem_b (conv3 + max-pool + 1x1) reads conv2's fp16
      // output whatever produced it.
      const int stem_a_max = getenv("DV_NO_STEM_A_WIDE") == nullptr ? dv::kStemA_MaxChannels : 8;
      if (ops[0].first_u8 && desc.channels <= stem_a_max && ops[3].pool_in && ops[3].cout <= 96) {
        ops[0].stem_a = true;
        buffers[ops[0].out_buf] = {1, 1, 32, 0};  // conv1 output: LDS only
      }
      if (ops[3].pool_in && ops[3].cout <= 96 && (ops[0].stem_a || getenv("DV_NO_STEM_B_ALONE") == nullptr)) {
        ops[2].stem_b = true;
        buffers[ops[2].out_buf] = {1, 1, 64, 0};  // conv3 output: LDS only
      }
    }
    x = conv(x, 192, 3, 3, 1, false);
    blank_conv4_op = static_cast<int>(ops.size()) - 1;
    // The stem's second max-pool has ONE consumer launch -- mixed0's four 1x1 heads, grouped
    // (the pooled branch projects before it averages) -- so it is taken on the fly there
    // (conv_pool1x1_kernel) and the pooled tensor is never written.  DV_NO_POOL2_FUSE keeps
    // the separate max-pool kernel.
    const bool fuse_pool2 = getenv("DV_NO_POOL2_FUSE") == nullptr && getenv("DV_NO_POOL_FUSE") == nullptr &&
                            getenv("DV_NO_GROUPING") == nullptr;
    const int pool2_ih = x.h, pool2_iw = x.w;
    // Round 4: the pool moves into its PRODUCER (conv_pool_resident_kernel): the 21 x 51 x 192 tensor
    // is never written, mixed0's heads read the pooled 10 x 25 x 192 tensor like any other block's.
    // DV_NO_POOL2_IN_CONV keeps the round-3 arrangement (pool on load in the heads).
    const bool pool_in_conv = fuse_pool2 && getenv("DV_NO_POOL2_IN_CONV") == nullptr && ops.back().nb == 3 &&
                              x.h >= 3 && x.w >= 3;
    if (pool_in_conv) {
      x.h = (x.h - 3) / 2 + 1;
      x.w = (x.w - 3) / 2 + 1;
      ops.back().pool_out = true;
      buffers[x.buf] = {x.h, x.w, x.c, 0};
    } else if (fuse_pool2) {
      x.h = (x.h - 3) / 2 + 1;
      x.w = (x.w - 3) / 2 + 1;
    } else {
      x = pool(kOpMaxPool, x);
    }
    // Everything up to here is the "stem": big feature maps (0.2-0.7 MB per
    // example each).  It runs in sub-batches of stem_sub_batch() examples over
    // small, reused buffers so that every producer->consumer hand-off stays in
    // the 256 MB Infinity Cache instead of streaming through HBM; only the
    // 96 KB/example stem output is written at full-batch width.
    stem_ops_end = static_cast<int>(ops.size());
    stem_out_buf = x.buf;
    for (int pool_ch : {32, 64, 64}) {  // mixed0..2
      const int out = new_buffer(x.h, x.w, 64 + 64 + 96 + pool_ch);
      conv(x, 64, 1, 1, 1, true, out, 0);
      TensorRef b5 = conv(x, 48, 1, 1);
      conv(b5, 64, 5, 5, 1, true, out, 64);
      TensorRef b3 = conv(x, 64, 1, 1);
      b3 = conv(b3, 96, 3, 3);
      conv(b3, 96, 3, 3, 1, true, out, 128);
      pooled_projection(x, pool_ch, out, 224);
      if (fuse_pool2 && !pool_in_conv && x.buf == stem_out_buf) {  // mixed0: its 1x1 heads pool their input
        for (size_t k = stem_ops_end; k < ops.size(); ++k) {
          if (ops[k].type == kOpConv && ops[k].in_buf == x.buf) {
            ops[k].pool_in = true;
            ops[k].ih = pool2_ih;
            ops[k].iw = pool2_iw;
          }
        }
      }
      x = full(out);
    }
    {  // mixed3
      const int oh = (x.h - 3) / 2 + 1, ow = (x.w - 3) / 2 + 1;
      const int out = new_buffer(oh, ow, 384 + 96 + x.c);
      conv(x, 384, 3, 3, 2, false, out, 0);
      TensorRef b = conv(x, 64, 1, 1);
      b = conv(b, 96, 3, 3);
      conv(b, 96, 3, 3, 2, false, out, 384);
      pool(kOpMaxPool, x, out, 480);
      x = full(out);
    }
    wide_stage = true;   // precise mode: the tensors created from here on (17x17 and 8x8 stages) are hi + lo
    for (int c7 : {128, 160, 160, 192}) {  // mixed4..7
      const int out = new_buffer(x.h, x.w, 768);
      conv(x, 192, 1, 1, 1, true, out, 0);
      TensorRef b = conv(x, c7, 1, 1);
      b = conv(b, c7, 1, 7);
      conv(b, 192, 7, 1, 1, true, out, 192);
      TensorRef d = conv(x, c7, 1, 1);
      d = conv(d, c7, 7, 1);
      d = conv(d, c7, 1, 7);
      d = conv(d, c7, 7, 1);
      conv(d, 192, 1, 7, 1, true, out, 384);
      pooled_projection(x, 192, out, 576);
      x = full(out);
    }
    {  // mixed8
      const int oh = (x.h - 3) / 2 + 1, ow = (x.w - 3) / 2 + 1;
      const int out = new_buffer(oh, ow, 320 + 192 + x.c);
      TensorRef b = conv(x, 192, 1, 1);
      conv(b, 320, 3, 3, 2, false, out, 0);
      TensorRef d = conv(x, 192, 1, 1);
      d = conv(d, 192, 1, 7);
      d = conv(d, 192, 7, 1);
      conv(d, 192, 3, 3, 2, false, out, 320);
      pool(kOpMaxPool, x, out, 512);
      x = full(out);
    }
    for (int i = 0; i < 2; ++i) {  // mixed9, mixed10
      const int out = new_buffer(x.h, x.w, 2048);
      conv(x, 320, 1, 1, 1, true, out, 0);
      TensorRef b = conv(x, 384, 1, 1);
      conv(b, 384, 1, 3, 1, true, out, 320);
      conv(b, 384, 3, 1, 1, true, out, 704);
      TensorRef d = conv(x, 448, 1, 1);
      d = conv(d, 384, 3, 3);
      conv(d, 384, 1, 3, 1, true, out, 1088);
      conv(d, 384, 3, 1, 1, true, out, 1472);
      pooled_projection(x, 192, out, 1856);
      x = full(out);
    }
    feat_buf = x.buf;
    buffers[feat_buf].f32 = true;            // the global pool reads float32
    buffers[feat_buf].wide = false;
    feat_p = x.h * x.w;
    feat_c = x.c;
    group_siblings();
    for (const Op& op : ops) {  // zero halo wide enough for every consumer
      int need = 0;
      if (op.type == kOpConv) need = std::max(op.pad_h, op.pad_w);
      if (op.type == kOpAvgPool) need = 1;  // avgpool3s1_kernel reads its taps unconditionally
      buffers[op.in_buf].halo = std::max(buffers[op.in_buf].halo, need);
    }
    choose_chains();
    if (getenv("DV_CHAIN_KEEP_HALO") == nullptr) {
      // A fused chain DMAs the INTERIOR of its input into LDS and handles the map border itself
      // (tap masks, the zero piece): its input tensor needs no halo in HBM.  Without one the rows
      // of a map are contiguous (a 4x12 map plane is 768 bytes = six whole 128-byte lines), so the
      // 1x1 head that produces the tensor stores whole lines instead of 192-byte row segments that
      // start mid-line, and the chain's input DMA is one run per plane.
      for (BufferDesc& b : buffers) b.halo = 0;
      for (const Op& op : ops) {
        if (op.in_chain) continue;
        int need = 0;
        if (op.type == kOpConv && op.chain_len == 0) need = std::max(op.pad_h, op.pad_w);
        if (op.type == kOpAvgPool) need = 1;
        buffers[op.in_buf].halo = std::max(buffers[op.in_buf].halo, need);
      }
    }
    choose_imgconv();
    choose_band();
    choose_split();
    choose_side_pool();
    choose_avg_epilogue();
    for (size_t i = 0; i < ops.size(); ++i) {  // packed-weight image per LAUNCH (after grouping)
      Op& op = ops[i];
      if (op.type != kOpConv) continue;
      int subs = 0;
      for (int gi = 0; gi <= op.group_followers; ++gi) subs += (ops[i + gi].cout + 31) / 32;
      const int n_tiles = (subs + op.nb - 1) / op.nb;
      for (int gi = 0; gi <= op.group_followers; ++gi) ops[i + gi].w_off = packed_halfs;
      packed_halfs += op.first_u8 ? static_cast<size_t>(kFirstMaxChunks) * 32 * kChunk
                      : (op.chain_len > 0 || op.in_chain)
                          ? static_cast<size_t>(op.n_chunks) * 2 * ((op.cout + 31) / 32 * 32) * 8
                      : op.v2     ? static_cast<size_t>(op.v2_tiles) * op.v2_steps *
                                        dv::imgconv_wslab_halfs(op.kh, op.kw, op.nb)
                                  : static_cast<size_t>(op.band ? op.band : 1) * n_tiles * op.n_steps *
                                        kSlabChunks * (op.nb * 32) * kChunk;
      i += op.group_followers;
    }
    layers.push_back({1, 1, feat_c, desc.num_classes, n_params});
    n_params += static_cast<int64_t>(feat_c) * desc.num_classes + desc.num_classes;
  }
};

namespace {

// 192-cout tiles (NB = 6), one pixel fragment per wave: the pixel operand -- the texture-
// addresser path that bounds the other shapes (HISTORY.md 7) -- is fetched once for all 192
// couts instead of once per 96-cout tile.  Weight slabs of 4 chunks keep two blocks per CU.
void launch_conv6(const ConvArgs& a, hipStream_t stream) {
  const long rows = a.band ? a.band : 1;
  const long row_px = a.band ? static_cast<long>(a.N) * a.OW : a.M;
  const long blocks = rows * ((row_px + 127) / 128) * a.n_tiles;
  constexpr size_t lds = static_cast<size_t>(2) * 4 * 192 * kChunk * 2;
  if (a.wide_in) {
    hipLaunchKernelGGL((conv_mfma_kernel<6, 1, 2, 4, 4, false, false, false, true>), dim3(static_cast<unsigned>(blocks)),
                       dim3(kConvThreads), lds, stream, a);
    return;
  }
  hipLaunchKernelGGL((conv_mfma_kernel<6, 1, 2, 4>), dim3(static_cast<unsigned>(blocks)), dim3(kConvThreads), lds,
                     stream, a);
}

template <int NB>
void launch_conv(const ConvArgs& a, hipStream_t stream) {
  const int n_tiles = a.n_tiles;
  // pixel blocks of `px` pixels: over all N*OH*OW pixels, or per output row in row-band mode
  const long rows = a.band ? a.band : 1;
  const long row_px = a.band ? static_cast<long>(a.N) * a.OW : a.M;
  auto blocks = [&](int px) { return rows * ((row_px + px - 1) / px) * n_tiles; };
  // Two pixel tiles per wave halve the LDS weight traffic per MFMA; fall back
  // to one when that would leave CUs without a block.
  const long blocks2 = blocks(256);
  if constexpr (NB == 4) {
    if (a.side_pool_out != nullptr) {   // its own instantiations: the side pool costs registers the other launches keep
      if (blocks2 >= 512) {
        hipLaunchKernelGGL((conv_mfma_kernel<NB, 2, 2, kSlabChunks, 4, false, true>), dim3(static_cast<unsigned>(blocks2)),
                           dim3(kConvThreads), conv_lds_bytes<NB>(), stream, a);
      } else {
        hipLaunchKernelGGL((conv_mfma_kernel<NB, 1, 2, kSlabChunks, 4, false, true>), dim3(static_cast<unsigned>(blocks(128))),
                           dim3(kConvThreads), conv_lds_bytes<NB>(), stream, a);
      }
      return;
    }
  }
  if (a.wide_in) {   // precise mode: hi + lo pixel fragments per K chunk (conv_slab_wide); the two usual tile shapes
    if constexpr (NB >= 2 && NB <= 4) {
      if constexpr (NB == 4) {
        if (a.tile_g > 0) {
          const long tiles = (static_cast<long>(a.N) + a.tile_g - 1) / a.tile_g * n_tiles;
          hipLaunchKernelGGL((conv_mfma_kernel<NB, 2, 2, kSlabChunks, 4, false, false, true, true>),
                             dim3(static_cast<unsigned>(tiles)), dim3(kConvThreads), conv_lds_bytes<NB>(), stream, a);
          return;
        }
      }
      if (blocks2 >= 512) {
        hipLaunchKernelGGL((conv_mfma_kernel<NB, 2, 2, kSlabChunks, 4, false, false, false, true>),
                           dim3(static_cast<unsigned>(blocks2)), dim3(kConvThreads), conv_lds_bytes<NB>(), stream, a);
      } else {
        hipLaunchKernelGGL((conv_mfma_kernel<NB, 1, 2, kSlabChunks, 4, false, false, false, true>),
                           dim3(static_cast<unsigned>(blocks(128))), dim3(kConvThreads), conv_lds_bytes<NB>(), stream, a);
      }
      return;
    }
  }
  if constexpr (NB == 4) {
    if (a.tile_g > 0) {   // image-aligned 256-pixel tiles, pooled projection averaged in the epilogue
      const long tiles = (static_cast<long>(a.N) + a.tile_g - 1) / a.tile_g * n_tiles;
      hipLaunchKernelGGL((conv_mfma_kernel<NB, 2, 2, kSlabChunks, 4, false, false, true>),
                         dim3(static_cast<unsigned>(tiles)), dim3(kConvThreads), conv_lds_bytes<NB>(), stream, a);
      return;
    }
  }
  if (a.split) {  // W_hi + W_lo images: the same two tile shapes, SPLIT slab code
    static const long pt2_min = getenv("DV_SPLIT_PT2_MIN") ? atol(getenv("DV_SPLIT_PT2_MIN")) : 256;  // split layers carry twice the weight bytes per pixel: two fragments per wave from 256 blocks up (1x3 / 3x1 / 3x3 of mixed9-10: -12...-15 %, tools/r4_run.sh ab:DV_SPLIT_PT2_MIN=256)
    if (blocks2 >= pt2_min) {
      hipLaunchKernelGGL((conv_mfma_kernel<NB, 2, 2, kSlabChunks, 4, true>), dim3(static_cast<unsigned>(blocks2)),
                         dim3(kConvThreads), conv_lds_bytes<NB>(), stream, a);
    } else {
      hipLaunchKernelGGL((conv_mfma_kernel<NB, 1, 2, kSlabChunks, 4, true>), dim3(static_cast<unsigned>(blocks(128))),
                         dim3(kConvThreads), conv_lds_bytes<NB>(), stream, a);
    }
    return;
  }
  // (Round 6, measured and removed: <4,4> -- 128 pixels x 128 couts per wave, 16 accumulators in AGPRs, one workgroup
  // per CU, half the weight-slab bytes per MFMA: 17x17 heads 455 -> 490 us, 524 -> 568; 35x35 heads 449 -> 528; the
  // step 465.6 -> 449.1 K candidates/s with the pools unfused on both sides.  profiles/r06_experiments.txt.)
  static const int force_pt = getenv("DV_CONV_PT") ? atoi(getenv("DV_CONV_PT")) : 0;  // tuning knob
  // Four pixel tiles per wave where the accumulators still leave two blocks per CU and
  // K is long enough to amortise the wider prologue: measured -6 % on the 32-cout stem
  // 3x3 and -7 % on the 5x5s; +11 % (slower) on <2,4> 3x3 and 1x1 layers.
  static const bool no_pt4 = getenv("DV_NO_PT4") != nullptr;  // tuning knob
  if constexpr (NB <= 2) {
    if (!no_pt4 && !force_pt && blocks2 >= 4096 && (NB == 1 || a.KH * a.KW >= 25)) {
      hipLaunchKernelGGL((conv_mfma_kernel<NB, 4>), dim3(static_cast<unsigned>(blocks(512))),
                         dim3(kConvThreads), conv_lds_bytes<NB>(), stream, a);
      return;
    }
  }
  // tuning experiments (HISTORY.md 7): 4-chunk weight slabs; 8-wave blocks of 512 pixels
  static const bool slab4 = getenv("DV_CONV_SLAB4") != nullptr;
  static const bool w8 = getenv("DV_CONV_W8") != nullptr;
  if constexpr (NB >= 3 && NB <= 4) {
    if (slab4 && blocks2 >= 512) {
      hipLaunchKernelGGL((conv_mfma_kernel<NB, 2, 2, 4>), dim3(static_cast<unsigned>(blocks2)),
                         dim3(kConvThreads), static_cast<size_t>(2) * 4 * NB * 32 * kChunk * 2, stream, a);
      return;
    }
    if (w8 && blocks2 >= 1024) {
      hipLaunchKernelGGL((conv_mfma_kernel<NB, 2, 1, kSlabChunks, 8>), dim3(static_cast<unsigned>(blocks(512))),
                         dim3(512), conv_lds_bytes<NB>(), stream, a);
      return;
    }
  }
  if (force_pt ? force_pt == 2 : blocks2 >= 512) {
    if constexpr (NB == 4) {
      // <4,2> compiled for two blocks per CU (249 VGPRs, no spills) instead of one with
      // accumulators in AGPRs (284): a second wave per SIMD covers the other's waits --
      // measured -12...-26 % on every nb4 layer (DV_CONV42_BLOCKS=1 restores one block).
      static const bool two = getenv("DV_CONV42_BLOCKS") == nullptr || atoi(getenv("DV_CONV42_BLOCKS")) != 1;
      if (two) {
        hipLaunchKernelGGL((conv_mfma_kernel<NB, 2, 2>), dim3(static_cast<unsigned>(blocks2)),
                           dim3(kConvThreads), conv_lds_bytes<NB>(), stream, a);
        return;
      }
    }
    hipLaunchKernelGGL((conv_mfma_kernel<NB, 2>), dim3(static_cast<unsigned>(blocks2)),
                       dim3(kConvThreads), conv_lds_bytes<NB>(), stream, a);
  } else {
    hipLaunchKernelGGL((conv_mfma_kernel<NB, 1>), dim3(static_cast<unsigned>(blocks(128))),
                       dim3(kConvThreads), conv_lds_bytes<NB>(), stream, a);
  }
}

// Runs ops [first, last) on `n` examples.  `out_example_off` shifts the output
// pointer of ops that write `shifted_buf` (the stem's full-batch output).
// DV_OP_TRACE=1: per-launch table (ms, TFLOP/s, activation GB/s) on stderr after
// every eager forward -- the per-layer view rocprofv3's per-kernel-name stats cannot give.
struct OpTrace {
  hipEvent_t a, b;
  std::string label;
  double flops, bytes;
};
std::vector<OpTrace>* g_trace = nullptr;

struct TraceScope {
  hipStream_t stream;
  bool on;
  TraceScope(hipStream_t s, std::string label, double flops, double bytes) : stream(s), on(g_trace != nullptr) {
    if (!on) return;
    OpTrace t{nullptr, nullptr, std::move(label), flops, bytes};
    (void)hipEventCreate(&t.a);
    (void)hipEventCreate(&t.b);
    (void)hipEventRecord(t.a, stream);
    g_trace->push_back(std::move(t));
  }
  ~TraceScope() {
    if (on) (void)hipEventRecord(g_trace->back().b, stream);
  }
};

void dump_trace(hipStream_t stream) {
  (void)hipStreamSynchronize(stream);
  double tot = 0;
  for (OpTrace& t : *g_trace) {
    float ms = 0;
    (void)hipEventElapsedTime(&ms, t.a, t.b);
    tot += ms;
    fprintf(stderr, "[dv-op] %-58s %8.1f us %7.1f TF/s %7.0f GB/s\n", t.label.c_str(), ms * 1e3,
            t.flops / (ms * 1e-3) / 1e12, t.bytes / (ms * 1e-3) / 1e9);
    (void)hipEventDestroy(t.a);
    (void)hipEventDestroy(t.b);
  }
  fprintf(stderr, "[dv-op] total %.1f us\n", tot * 1e3);
  g_trace->clear();
}

// conv_resident_kernel applies when the whole 96-cout tile fits the LDS next to nothing else,
// the launch has enough pixels to give every wave of a persistent grid several tiles, and no
// special mode is on.  DV_RESIDENT=0 keeps conv_mfma_kernel everywhere; DV_RESIDENT=2 widens it
// from the 3x3 80->192 to every eligible 96-cout-tile layer (tuning).
bool resident_ok(const dv_model* m, const Op& op, const ConvArgs& a) {
  const char* env = getenv("DV_RESIDENT");   // read per launch set-up (tests toggle it between models)
  const int mode = env ? atoi(env) : 1;
  if (mode == 0 || op.nb != 3 || op.band || op.v2 || op.pool_in || op.split || a.blank_row != nullptr || a.wide_in) return false;
  const size_t lds = static_cast<size_t>(a.n_slabs) * kSlabChunks * 3 * 32 * kChunk * 2;
  if (lds > 150 * 1024 || m->n_cus < 8 * a.n_tiles) return false;
  static const long min_tiles = getenv("DV_RESIDENT_MIN_TILES") ? atol(getenv("DV_RESIDENT_MIN_TILES")) : 4;   // tuning knob
  if (static_cast<long>(a.M) < static_cast<long>(m->n_cus) * 8 * 64 * min_tiles) return false;
  if (mode >= 2) return true;
  return m->blank_conv4_op >= 0 && &op == &m->ops[m->blank_conv4_op];
}

int run_ops(dv_model* m, int first, int last, int n, hipStream_t stream,
            int shifted_buf = -1, int out_example_off = 0, size_t images_off = 0) {
  std::vector<char> side_pooled(m->ops.size(), 0);   // max-pools a convolution of this pass has taken on the side
  for (int oi = first; oi < last; ++oi) {
    const Op& op = m->ops[oi];
    if (side_pooled[oi]) continue;
    if (op.type == kOpAvgPool && op.avg_partner >= 0 && m->ops[op.avg_partner].avg_partner == oi &&
        op.avg_partner >= first) {
      continue;   // averaged in the epilogue of the launch that holds its 1x1 (choose_avg_epilogue)
    }
    const BufferDesc& ob = m->buffers[op.out_buf];
    const size_t out_shift_halfs =
        op.out_buf == shifted_buf ? static_cast<size_t>(out_example_off) * ob.bytes_per_example() / 2
                                  : 0;
    if (op.type == kOpConv && op.stem_a) {
      const Op& c2 = m->ops[oi + 1];
      const BufferDesc& o2 = m->buffers[c2.out_buf];
      dv::StemAArgs a{};
      a.in = nullptr;
      a.in_ind = &static_cast<const ExtPtrs*>(m->d_ext.ptr)->images;
      a.in_off = images_off;
      a.w1 = static_cast<const _Float16*>(m->d_w.ptr) + op.w_off;
      a.w2 = static_cast<const _Float16*>(m->d_w.ptr) + c2.w_off;
      a.shift1 = static_cast<const float*>(m->d_shift.ptr) + op.shift_off;
      a.shift2 = static_cast<const float*>(m->d_shift.ptr) + c2.shift_off;
      a.out = static_cast<_Float16*>(m->dbuf[c2.out_buf].ptr);
      const TensorGeom g = o2.geom();
      a.og = dv::C8Geom{g.h, g.w, g.halo, g.hp, g.wp, g.groups};
      a.N = n;
      a.H = op.ih;
      a.W = op.iw;
      a.C = op.cin_real;
      a.OH1 = op.oh;
      a.OW1 = op.ow;
      a.OH2 = c2.oh;
      a.OW2 = c2.ow;
      a.tiles_y = (c2.oh + dv::kStemA_TH - 1) / dv::kStemA_TH;
      a.tiles_x = (c2.ow + dv::kStemA_TW - 1) / dv::kStemA_TW;
      a.total_tiles = n * a.tiles_y * a.tiles_x;
      a.in_bytes = static_cast<unsigned>(static_cast<size_t>(n) * op.ih * op.iw * op.cin_real);
      if (m->blank_on()) {
        a.blank_thr = static_cast<const int*>(m->d_blank_thr.ptr) + 1 * m->desc.max_batch;
        a.blank_src = static_cast<const _Float16*>(m->d_blank_c2.ptr);
        a.blank_need = static_cast<const int*>(m->d_blank_thr.ptr) + 5 * m->desc.max_batch;
      }
      TraceScope tr(stream, std::string(m->blank_on() ? "[blank tiles copied] " : "") + "stem_a conv3x3s2 " + std::to_string(op.cin_real) + "->32 + conv3x3 32->32 (fused)",
                    2.0 * n * (static_cast<double>(op.oh) * op.ow * op.kh * op.kw * op.cin_real * op.cout +
                               static_cast<double>(c2.oh) * c2.ow * 9 * 32 * 32),
                    static_cast<double>(n) * (op.ih * op.iw * op.cin_real + 2.0 * c2.oh * c2.ow * 32));
      dv::ProfileScope prof(dv::kProfConv, stream);
      dv::launch_stem_a(a, m->stem_a_grid, stream);
      oi += 1;
    } else if (op.type == kOpConv && op.stem_b) {
      const Op& c4 = m->ops[oi + 1];
      const BufferDesc& ib = m->buffers[op.in_buf];
      const BufferDesc& o4 = m->buffers[c4.out_buf];
      dv::StemBArgs a{};
      a.in = static_cast<const _Float16*>(m->dbuf[op.in_buf].ptr);
      a.w3 = static_cast<const _Float16*>(m->d_w.ptr) + op.w_off;
      a.w4 = static_cast<const _Float16*>(m->d_w.ptr) + c4.w_off;
      a.shift3 = static_cast<const float*>(m->d_shift.ptr) + op.shift_off;
      a.shift4 = static_cast<const float*>(m->d_shift.ptr) + c4.shift_off;
      a.out = static_cast<_Float16*>(m->dbuf[c4.out_buf].ptr);
      const TensorGeom gi = ib.geom(), go = o4.geom();
      a.ig = dv::C8Geom{gi.h, gi.w, gi.halo, gi.hp, gi.wp, gi.groups};
      a.og = dv::C8Geom{go.h, go.w, go.halo, go.hp, go.wp, go.groups};
      a.N = n;
      a.OH3 = op.oh;
      a.OW3 = op.ow;
      a.PH = c4.oh;
      a.PW = c4.ow;
      a.Cout4 = c4.cout;
      a.tiles_y = (c4.oh + dv::kStemB_PH - 1) / dv::kStemB_PH;
      a.tiles_x = (c4.ow + dv::kStemB_PW - 1) / dv::kStemB_PW;
      a.total_tiles = n * a.tiles_y * a.tiles_x;
      a.in_bytes = static_cast<size_t>(n) * ib.bytes_per_example();
      a.in_img_bytes = static_cast<unsigned>(ib.bytes_per_example());
      if (m->blank_on()) {
        a.blank_thr = static_cast<const int*>(m->d_blank_thr.ptr) + 2 * m->desc.max_batch;
        a.blank_src = static_cast<const _Float16*>(m->d_blank_b.ptr);
        a.blank_need = static_cast<const int*>(m->d_blank_thr.ptr) + 6 * m->desc.max_batch;
      }
      TraceScope tr(stream, std::string(m->blank_on() ? "[blank tiles copied] " : "") + "stem_b conv3x3 32->64 + maxpool3s2 + conv1x1 64->" + std::to_string(c4.cout) + " (fused)",
                    2.0 * n * (static_cast<double>(op.oh) * op.ow * 9 * 32 * 64 +
                               static_cast<double>(c4.oh) * c4.ow * 64 * c4.cout),
                    2.0 * n * (static_cast<double>(op.ih) * op.iw * 32 + static_cast<double>(c4.oh) * c4.ow * c4.cout));
      dv::ProfileScope prof(dv::kProfConv, stream);
      static const bool stem_prof = getenv("DV_STEM_PROF") != nullptr;  // tuning aid, eager only
      if (stem_prof && g_trace != nullptr) {
        const size_t words = static_cast<size_t>(m->stem_b_grid) * 16;
        unsigned long long* d = nullptr;
        if (hipMalloc(&d, words * 8) == hipSuccess) {
          (void)hipMemsetAsync(d, 0, words * 8, stream);
          a.prof = d;
          dv::launch_stem_b(a, m->stem_b_grid, stream);
          std::vector<unsigned long long> h(words);
          (void)hipStreamSynchronize(stream);
          (void)hipMemcpy(h.data(), d, words * 8, hipMemcpyDeviceToHost);
          (void)hipFree(d);
          for (int w = 0; w < 2; ++w) {
            double sum[7] = {0, 0, 0, 0, 0, 0, 0};
            for (int b = 0; b < m->stem_b_grid; ++b)
              for (int i = 0; i < 7; ++i) sum[i] += static_cast<double>(h[(b * 2 + w) * 8 + i]);
            const double tiles = static_cast<double>(a.total_tiles);
            fprintf(stderr, "[dv-stem-b wave %d] cycles/tile: issue %.0f conv3 %.0f dma-wait %.0f barrierA %.0f pool %.0f "
                            "barrierB %.0f conv1x1+store %.0f\n", w ? 7 : 0, sum[0] / tiles, sum[1] / tiles,
                    sum[6] / tiles, sum[2] / tiles, sum[3] / tiles, sum[4] / tiles, sum[5] / tiles);
          }
          {  // placement: which workgroups share a CU, their TG slots, and the phase sums of the two classes
            std::map<unsigned long long, std::vector<int>> where;
            for (int b = 0; b < m->stem_b_grid; ++b) {
              const unsigned long long v = h[(b * 2) * 8 + 7];
              const unsigned hw = static_cast<unsigned>(v), xcc = static_cast<unsigned>(v >> 32);
              where[(static_cast<unsigned long long>(xcc) << 16) | ((hw >> 8) & 0xffu)].push_back(b);
            }
            int pairs = 0, same_parity = 0, apart256 = 0;
            for (const auto& kv : where) {
              if (kv.second.size() != 2) continue;
              ++pairs;
              const unsigned t0 = (static_cast<unsigned>(h[(kv.second[0] * 2) * 8 + 7]) >> 16) & 15u;
              const unsigned t1 = (static_cast<unsigned>(h[(kv.second[1] * 2) * 8 + 7]) >> 16) & 15u;
              same_parity += (t0 & 1u) == (t1 & 1u);
              apart256 += kv.second[1] - kv.second[0] == 256;
            }
            fprintf(stderr, "[dv-stem-b placement] %zu places for %d workgroups; %d pairs, %d with TG slots of EQUAL parity, "
                            "%d pairs are blocks b / b+256\n", where.size(), m->stem_b_grid, pairs, same_parity, apart256);
            for (int cls = 0; cls < 2; ++cls) {
              double sum[7] = {0, 0, 0, 0, 0, 0, 0};
              int nb = 0;
              for (int b = 0; b < m->stem_b_grid; ++b) {
                const unsigned tg = (static_cast<unsigned>(h[(b * 2) * 8 + 7]) >> 16) & 15u;
                if (static_cast<int>(tg & 1u) != cls) continue;
                ++nb;
                for (int i = 0; i < 7; ++i) sum[i] += static_cast<double>(h[(b * 2) * 8 + i]);
              }
              double tot = 0;
              for (double v : sum) tot += v;
              fprintf(stderr, "[dv-stem-b TG parity %d] %d workgroups, share of wave-0 cycles: issue %.3f conv3 %.3f dma-wait %.3f "
                              "barrierA %.3f pool %.3f barrierB %.3f conv1x1+store %.3f; cycles per workgroup %.0f\n", cls, nb,
                      sum[0] / tot, sum[1] / tot, sum[6] / tot, sum[2] / tot, sum[3] / tot, sum[4] / tot, sum[5] / tot,
                      nb ? tot / nb : 0.0);
            }
          }
        }
      } else {
        dv::launch_stem_b(a, m->stem_b_grid, stream);
      }
      oi += 1;
    } else if (op.type == kOpConv && op.chain_len > 0) {
      const Op& last = m->ops[oi + op.chain_len - 1];
      const BufferDesc& ib = m->buffers[op.in_buf];
      const BufferDesc& lob = m->buffers[last.out_buf];
      dv::ChainArgs a{};
      a.in = static_cast<const _Float16*>(m->dbuf[op.in_buf].ptr);
      a.ig = ib.geom();
      a.in_img_bytes = static_cast<unsigned>(ib.bytes_per_example());
      a.N = n;
      a.G = op.chain_g;
      a.tpx = op.chain_tpx;
      a.h = op.oh;
      a.w = op.ow;
      a.n_tiles = (n + a.G - 1) / a.G;
      a.n_layers = op.chain_len;
      double tr_flops = 0;
      std::string tr_label = "chain";
      size_t act = 0, slot = 0;
      for (int k = 0; k < op.chain_len; ++k) {
        const Op& o = m->ops[oi + k];
        dv::ChainLayer& cl = a.L[k];
        cl.w = static_cast<const _Float16*>(m->d_w.ptr) + o.w_off;
        cl.shift = static_cast<const float*>(m->d_shift.ptr) + o.shift_off;
        cl.n_chunks = o.cin / kChunk;
        cl.cout = o.cout;
        cl.cout_pad = (o.cout + 31) / 32 * 32;
        cl.kh = o.kh;
        cl.kw = o.kw;
        cl.slab_bytes = static_cast<unsigned>(o.kh * o.kw * 2 * cl.cout_pad * 16);
        act = std::max(act, static_cast<size_t>(o.cin / 8) * op.chain_tpx * 16);
        slot = std::max(slot, static_cast<size_t>(cl.slab_bytes));
        tr_flops += 2.0 * n * o.oh * o.ow * o.kh * o.kw * o.cin * o.cout;
        tr_label += " " + std::to_string(o.kh) + "x" + std::to_string(o.kw) + ":" + std::to_string(o.cin) + "->" +
                    std::to_string(o.cout);
      }
      a.act_bytes = static_cast<unsigned>(act);
      a.slot_bytes = static_cast<unsigned>(slot);
      a.out = static_cast<_Float16*>(m->dbuf[last.out_buf].ptr);
      a.og = lob.geom();
      a.out_goff = last.out_coff / 8;
      tr_label += " @" + std::to_string(op.oh) + "x" + std::to_string(op.ow) + " [fused, G=" + std::to_string(a.G) + "]";
      TraceScope tr(stream, tr_label, tr_flops,
                    2.0 * n * op.oh * op.ow * (static_cast<double>(op.cin) + last.cout));
      dv::ProfileScope prof(dv::kProfConv, stream);
      static const bool chain_prof = getenv("DV_CHAIN_PROF") != nullptr;   // tuning aid, eager only
      if (chain_prof && g_trace != nullptr) {
        const int grid = std::min(a.n_tiles, m->n_cus);
        const size_t words = static_cast<size_t>(grid) * 4 * 8;
        unsigned long long* d = nullptr;
        if (hipMalloc(&d, words * 8) == hipSuccess) {
          (void)hipMemsetAsync(d, 0, words * 8, stream);
          a.prof = d;
          dv::launch_chain(a, m->n_cus, stream);
          std::vector<unsigned long long> h(words);
          (void)hipStreamSynchronize(stream);
          (void)hipMemcpy(h.data(), d, words * 8, hipMemcpyDeviceToHost);
          (void)hipFree(d);
          for (int w = 0; w < 4; ++w) {
            double sum[6] = {0, 0, 0, 0, 0, 0};
            for (int b = 0; b < grid; ++b)
              for (int i = 0; i < 6; ++i) sum[i] += static_cast<double>(h[(static_cast<size_t>(b) * 4 + w) * 8 + i]);
            const double tiles = static_cast<double>(a.n_tiles);
            fprintf(stderr, "[dv-chain wave %d] cycles/tile: chunk-barrier wait %.0f mfma steps %.0f layer-barrier wait %.0f "
                            "lds epilogue %.0f hbm epilogue %.0f set-up %.0f\n", w, sum[0] / tiles, sum[1] / tiles,
                    sum[2] / tiles, sum[3] / tiles, sum[4] / tiles, sum[5] / tiles);
          }
        }
      } else {
        dv::launch_chain(a, m->n_cus, stream);
      }
      oi += op.chain_len - 1;
    } else if (op.type == kOpConv && op.first_u8) {
      FirstConvArgs f{};
      f.ext = static_cast<const ExtPtrs*>(m->d_ext.ptr);
      f.in_off = images_off;
      f.w = static_cast<const _Float16*>(m->d_w.ptr) + op.w_off;
      f.shift = static_cast<const float*>(m->d_shift.ptr) + op.shift_off;
      f.out = static_cast<_Float16*>(m->dbuf[op.out_buf].ptr) + out_shift_halfs;
      f.og = ob.geom();
      f.N = n;
      f.H = op.ih;
      f.W = op.iw;
      f.C = op.cin_real;
      f.Cout = op.cout;
      f.OH = op.oh;
      f.OW = op.ow;
      f.KH = op.kh;
      f.KW = op.kw;
      f.stride = op.stride;
      f.M = n * op.oh * op.ow;
      f.n_chunks = op.n_chunks;
      f.wide = op.cin_real > 8 ? 1 : 0;
      f.in_bytes = static_cast<unsigned>(static_cast<size_t>(n) * op.ih * op.iw * op.cin_real);
      f.rcp_ow = 1.0f / static_cast<float>(op.ow);
      f.rcp_ohow = 1.0f / static_cast<float>(op.oh * op.ow);
      TraceScope tr(stream, "conv_first_u8 3x3 s2 " + std::to_string(op.cin_real) + "->" + std::to_string(op.cout),
                    2.0 * f.M * op.kh * op.kw * op.cin_real * op.cout,
                    static_cast<double>(n) * (op.ih * op.iw * op.cin_real + 2.0 * op.oh * op.ow * op.cout));
      dv::ProfileScope prof(dv::kProfConv, stream);
      // four pixel fragments per wave: 20 outstanding 12-byte loads per lane (+1.7 % end to end
      // over two on MI355X); DV_FIRST_PT2 restores the smaller tile for tuning.
      static const bool first4 = getenv("DV_FIRST_PT2") == nullptr;
      // wide inputs (9..16 channels): four fragments per wave as well -- 36 outstanding 12-byte loads per lane,
      // 236 VGPRs; hifi35 504.2 -> 516.6 K, ont50 369.4 -> 376.6 K candidates/s same box (DV_FIRST_WIDE_PT2 restores <2,9>)
      static const bool wide4 = getenv("DV_FIRST_WIDE_PT2") == nullptr;
      if (f.wide && wide4) {
        hipLaunchKernelGGL((conv_first_u8_kernel<4, 9>), dim3((f.M + 511) / 512), dim3(kConvThreads), 0,
                           stream, f);
      } else if (f.wide) {   // nine one-tap chunks, all requested before the first MFMA
        hipLaunchKernelGGL((conv_first_u8_kernel<2, 9>), dim3((f.M + 255) / 256), dim3(kConvThreads), 0,
                           stream, f);
      } else if (first4) {
        hipLaunchKernelGGL((conv_first_u8_kernel<4>), dim3((f.M + 511) / 512), dim3(kConvThreads), 0,
                           stream, f);
      } else {
        hipLaunchKernelGGL((conv_first_u8_kernel<2>), dim3((f.M + 255) / 256), dim3(kConvThreads), 0,
                           stream, f);
      }
    } else if (op.type == kOpConv) {
      ConvArgs a{};
      a.in = static_cast<const _Float16*>(m->dbuf[op.in_buf].ptr);
      const BufferDesc& ib = m->buffers[op.in_buf];
      a.ig = ib.geom();
      a.N = n;
      a.Cin = op.cin;
      a.OH = op.oh;
      a.OW = op.ow;
      static const int cu_pair = getenv("DV_CU_PAIR") ? atoi(getenv("DV_CU_PAIR")) : 0;
      a.cu_pair = cu_pair;
      a.band = op.band;
      a.KH = op.band ? op.band : op.kh;
      a.KW = op.kw;
      a.stride = op.stride;
      a.pad_h = op.pad_h;
      a.pad_w = op.pad_w;
      a.chunk_stride = static_cast<unsigned>(2 * a.ig.hp * a.ig.wp * 16);
      a.M = n * op.oh * op.ow;
      a.n_chunks = op.n_chunks;
      a.n_slabs = op.n_steps;
      a.split = op.split ? 1 : 0;
      a.split_tiles = op.split_tiles;
      a.wide_in = op.in_wide ? 1 : 0;
      a.lo_off = op.in_wide ? static_cast<unsigned>(ib.c / 8) * static_cast<unsigned>(a.ig.hp * a.ig.wp) * 16u : 0u;
      a.in_bytes = static_cast<size_t>(n) * ib.bytes_per_example();
      a.img_bytes = static_cast<unsigned>(ib.bytes_per_example());
      a.rcp_ow = 1.0f / static_cast<float>(op.ow);
      a.rcp_ohow = 1.0f / static_cast<float>(op.oh * op.ow);
      // this op + the sibling convs grouped behind it (same input, same geometry)
      int subs = 0;
      a.n_branches = 0;
      double tr_flops = 0, tr_bytes = static_cast<double>(n) * op.ih * op.iw * op.cin * 2.0;
      std::string tr_label = "conv " + std::to_string(op.kh) + "x" + std::to_string(op.kw) + " s" +
                             std::to_string(op.stride) + " " + std::to_string(op.cin) + "->";
      for (int gi = 0; gi <= op.group_followers; ++gi) {
        const Op& bo = m->ops[oi + gi];
        tr_flops += 2.0 * n * op.oh * op.ow * op.kh * op.kw * op.cin_real * bo.cout;
        tr_bytes += 2.0 * n * op.oh * op.ow * bo.cout;
        tr_label += (gi ? "+" : "") + std::to_string(bo.cout);
        const BufferDesc& bob = m->buffers[bo.out_buf];
        ConvBranch& br = a.br[a.n_branches++];
        br.shift = bo.raw ? nullptr
                          : static_cast<const float*>(m->d_shift.ptr) + bo.shift_off;
        br.out = static_cast<_Float16*>(m->dbuf[bo.out_buf].ptr) +
                 (bo.out_buf == shifted_buf
                      ? static_cast<size_t>(out_example_off) * bob.bytes_per_example() / 2
                      : 0);
        br.out32 = bob.f32 ? static_cast<float*>(m->dbuf[bo.out_buf].ptr) : nullptr;   // (never the stem's shifted buffer)
        br.og = bob.geom();
        br.lo_groups = bob.wide ? bob.c / 8 : 0;
        br.out_goff = bo.out_coff / 8;
        br.Cout = bo.cout;
        br.relu = bo.raw ? 0 : 1;
        br.sub0 = subs;
        subs += (bo.cout + 31) / 32;
        if (op.avg_tile_g > 0 && bo.avg_partner >= 0 && bo.avg_partner < last) {   // pooled in this launch's epilogue
          const Op& pl = m->ops[bo.avg_partner];
          const BufferDesc& pb = m->buffers[pl.out_buf];
          br.avgpool = 1;
          br.shift = static_cast<const float*>(m->d_shift.ptr) + pl.shift_off;
          br.relu = 1;
          br.out = static_cast<_Float16*>(m->dbuf[pl.out_buf].ptr) +
                   (pl.out_buf == shifted_buf ? static_cast<size_t>(out_example_off) * pb.bytes_per_example() / 2 : 0);
          br.out32 = pb.f32 ? static_cast<float*>(m->dbuf[pl.out_buf].ptr) : nullptr;
          br.og = pb.geom();
          br.lo_groups = pb.wide ? pb.c / 8 : 0;
          br.out_goff = pl.out_coff / 8;
          a.tile_g = op.avg_tile_g;
          a.tile_p = op.oh * op.ow;
          a.rcp_tile_p = 1.0f / static_cast<float>(a.tile_p);
        }
      }
      if (a.tile_g > 0) tr_label += " [+ avgpool3s1 in the epilogue, " + std::to_string(a.tile_g) + " maps per block]";
      a.w = static_cast<const _Float16*>(m->d_w.ptr) + op.w_off;
      const int tiles = (subs + op.nb - 1) / op.nb;
      a.n_tiles = tiles;
      if (m->blank_on() && m->blank_conv4_op >= 0 && &op == &m->ops[m->blank_conv4_op] &&
          a.n_branches == 1 && !op.band && !op.v2 && !op.pool_in) {
        a.blank_row = static_cast<const int*>(m->d_blank_thr.ptr) + (op.pool_out ? 4 : 3) * m->desc.max_batch;
        a.blank_src = static_cast<const _Float16*>(m->d_blank_conv4.ptr);
        tr_label += " [blank rows copied]";
      }
      if (m->blank_on() && oi == 1 && !m->ops[0].stem_a && a.n_branches == 1 && !op.band && !op.v2 && !op.pool_in &&
          !op.pool_out && !op.split && op.nb <= 4 && m->d_blank_c2.ptr != nullptr) {
        // inputs of 9..16 channels: conv2 runs per layer (conv_mfma_kernel) and skips like the fused stem_a does
        a.blank_row = static_cast<const int*>(m->d_blank_thr.ptr) + 1 * m->desc.max_batch;
        a.blank_src = static_cast<const _Float16*>(m->d_blank_c2.ptr);
        if (m->ops[2].stem_b) a.blank_need = static_cast<const int*>(m->d_blank_thr.ptr) + 5 * m->desc.max_batch;
        tr_label += " [blank rows copied]";
      }
      oi += op.group_followers;  // the followers ran in this launch
      tr_label += " @" + std::to_string(op.oh) + "x" + std::to_string(op.ow) + " nb" + std::to_string(op.nb) +
                  " tiles" + std::to_string(tiles);
      if (op.pool_in) tr_label += " <- maxpool3s2";
      if (op.in_wide) tr_label += " [hi+lo input: 2 MFMAs per weight fragment]";
      if (op.v2) tr_label += " [imgconv G=" + std::to_string(op.v2_g) + "]";
      if (op.band) tr_label += " [band: " + std::to_string(op.band) + " of " + std::to_string(op.kh) + " tap rows]";
      if (op.split) tr_label += " [split W: " + std::to_string(op.split_tiles) + " of " + std::to_string(tiles) + " tiles]";
      if (op.side_pool_partner > oi && op.side_pool_partner < last && op.nb == 4 && !resident_ok(m, op, a)) {
        const Op& pl = m->ops[op.side_pool_partner];
        a.side_pool_out = static_cast<_Float16*>(m->dbuf[pl.out_buf].ptr);
        a.side_pool_og = m->buffers[pl.out_buf].geom();
        a.side_pool_goff = pl.out_coff / 8;
        side_pooled[op.side_pool_partner] = 1;
        tr_label += " + maxpool3s2 on the side";
        tr_bytes += 2.0 * n * pl.oh * pl.ow * pl.cin;
      }
      const bool resident = !op.v2 && !op.pool_in && !op.pool_out && resident_ok(m, op, a);
      if (resident) tr_label += " [weights resident in LDS]";
      if (op.pool_out) tr_label += " [weights resident in LDS] -> maxpool3s2";
      TraceScope tr(stream, tr_label, tr_flops, tr_bytes);
      dv::ProfileScope prof(dv::kProfConv, stream);
      if (op.pool_out) {
        const size_t lds = static_cast<size_t>(a.n_slabs) * kSlabChunks * 3 * 32 * kChunk * 2;
        static const bool attr = [] {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pool_resident_kernel<3>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          return true;
        }();
        (void)attr;
        // one wave per fragment of 30 new positions of the (example, conv column) index; a persistent
        // grid of up to one block per CU, in whole sets of 8 blocks per cout tile
        const long frags = (static_cast<long>(n) * op.ow + 29) / 30;
        const int per_set = 8 * a.n_tiles;
        const long want = (frags + 8 * 8 - 1) / (8 * 8) * per_set;   // 8 waves x 8 blocks cover 64 fragments per set
        const int grid = static_cast<int>(std::max<long>(per_set, std::min<long>(want, m->n_cus / per_set * per_set)));
        hipLaunchKernelGGL((conv_pool_resident_kernel<3>), dim3(grid), dim3(512), lds, stream, a);
      } else if (op.v2) {
        dv::ImgConvArgs ia = m->imgconv_geometry(op, op.v2_g);
        ia.c = a;
        ia.n_img_tiles = (n + op.v2_g - 1) / op.v2_g;
        ia.n_cout_tiles = op.v2_tiles;
        dv::launch_imgconv(ia, op.nb, m->n_cus, stream);
      } else if (op.pool_in) {
        a.stride = 2;  // documentary: the window origin is (2 oh, 2 ow)
        const size_t lds = static_cast<size_t>(op.n_steps) * kSlabChunks * op.nb * 32 * kChunk * 2;
        const dim3 grid(static_cast<unsigned>(((a.M + 127) / 128) * a.n_tiles));
        if (op.nb == 7) {  // mixed0's heads: all 7 subtiles in one tile, every window pooled once
          static const bool attr = [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pool1x1_kernel<7, 8>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            return true;
          }();
          (void)attr;
          hipLaunchKernelGGL((conv_pool1x1_kernel<7, 8>), dim3(static_cast<unsigned>((a.M + 255) / 256)),
                             dim3(512), lds, stream, a);
        } else
        switch (op.nb) {
          case 1: hipLaunchKernelGGL((conv_pool1x1_kernel<1>), grid, dim3(kConvThreads), lds, stream, a); break;
          case 2: hipLaunchKernelGGL((conv_pool1x1_kernel<2>), grid, dim3(kConvThreads), lds, stream, a); break;
          case 3: hipLaunchKernelGGL((conv_pool1x1_kernel<3>), grid, dim3(kConvThreads), lds, stream, a); break;
          default: hipLaunchKernelGGL((conv_pool1x1_kernel<4>), grid, dim3(kConvThreads), lds, stream, a); break;
        }
      } else if (resident) {
        const size_t lds = static_cast<size_t>(a.n_slabs) * kSlabChunks * 3 * 32 * kChunk * 2;
        static const bool attr = [] {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_resident_kernel<3, 2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          return true;
        }();
        (void)attr;
        const int grid = m->n_cus / (8 * a.n_tiles) * (8 * a.n_tiles);
        hipLaunchKernelGGL((conv_resident_kernel<3, 2>), dim3(grid), dim3(512), lds, stream, a);
      } else
      switch (op.nb) {
        case 1: launch_conv<1>(a, stream); break;
        case 2: launch_conv<2>(a, stream); break;
        case 3: launch_conv<3>(a, stream); break;
        case 6: launch_conv6(a, stream); break;
        default: launch_conv<4>(a, stream); break;
      }
    } else {
      PoolArgs p{};
      p.in = static_cast<const _Float16*>(m->dbuf[op.in_buf].ptr);
      p.in32 = static_cast<const float*>(m->dbuf[op.in_buf].ptr);
      p.out = static_cast<_Float16*>(m->dbuf[op.out_buf].ptr) + out_shift_halfs;
      p.out32 = ob.f32 ? static_cast<float*>(m->dbuf[op.out_buf].ptr) : nullptr;
      p.lo_in_groups = m->buffers[op.in_buf].wide ? m->buffers[op.in_buf].c / 8 : 0;
      p.lo_out_groups = ob.wide ? ob.c / 8 : 0;
      p.ig = m->buffers[op.in_buf].geom();
      p.og = ob.geom();
      p.N = n;
      p.C = op.cin;
      p.OH = op.oh;
      p.OW = op.ow;
      p.out_goff = op.out_coff / 8;
      p.shift = op.pool_shift_relu
                    ? static_cast<const float*>(m->d_shift.ptr) + op.shift_off
                    : nullptr;
      const size_t total = static_cast<size_t>(n) * op.oh *
                           (op.type == kOpAvgPool ? (op.ow + 1) / 2 : op.ow) * (op.cin / 8);
      const dim3 grid(static_cast<unsigned>((total + 255) / 256));
      TraceScope tr(stream, std::string(op.type == kOpMaxPool ? "maxpool3s2 " : "avgpool3s1 ") +
                                std::to_string(op.cin) + " @" + std::to_string(op.oh) + "x" + std::to_string(op.ow),
                    0.0, 2.0 * n * op.cin * (static_cast<double>(op.ih) * op.iw + op.oh * op.ow));
      dv::ProfileScope prof(dv::kProfOther, stream);
      if (op.type == kOpMaxPool) {
        if (m->buffers[op.in_buf].f32 || ob.f32) return dv::fail(DV_ERR_UNSUPPORTED, "max-pool of a float32 tensor");
        hipLaunchKernelGGL(maxpool3s2_kernel, grid, dim3(256), 0, stream, p);
      } else {
        if (!m->buffers[op.in_buf].f32) return dv::fail(DV_ERR_UNSUPPORTED, "average pool of an fp16 tensor");
        hipLaunchKernelGGL(avgpool3s1_kernel, grid, dim3(256), 0, stream, p);
      }
    }
  }
  DV_HIP_CHECK(hipGetLastError());
  return DV_OK;
}

}  // namespace

extern "C" {

int dv_model_create(const dv_model_desc* desc, int device, dv_model** out) {
  if (!desc || !out) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_model_create: null");
  if (desc->channels < 1 || desc->channels > 16 || desc->num_classes < 1 ||
      desc->num_classes > 8 || desc->max_batch < 1 || desc->max_batch > 8192 ||
      desc->height < 75 ||
      desc->width < 75) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT,
                    "dv_model_create: unsupported shape (need H,W >= 75, C <= 16, max_batch <= 8192)");
  }
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
    return dv::fail(DV_ERR_NO_DEVICE, "no HIP device: libdvhip has no CPU fallback");
  }
  if (device < 0 || device >= n_dev) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "bad device ordinal");
  }
  DV_HIP_CHECK(hipSetDevice(device));
  std::unique_ptr<dv_model> m(new dv_model());
  m->device = device;
  m->desc = *desc;
  // Precise mode: on by default for inputs of more than 8 channels (the long-read models, whose deeper pile-ups do not
  // hold 1e-3 on every weight seed with fp16 activations: DESIGN.md 6), off for the short-read shapes; DV_PRECISE=0 / 1
  // overrides either way.
  m->precise = getenv("DV_PRECISE") != nullptr ? atoi(getenv("DV_PRECISE")) != 0 : desc->channels > 8;
  m->build();
  m->stem_a_grid = dv::stem_a_blocks(device);
  m->stem_b_grid = dv::stem_b_blocks(device);
  m->n_cus = dv::stem_a_blocks(device) / 2;   // stem_a runs two workgroups per CU
  // 32-bit index ranges of the kernels at max_batch (see conv_mfma_kernel's prologue)
  for (const Op& op : m->ops) {
    const BufferDesc& ob = m->buffers[op.out_buf];
    const double pieces = static_cast<double>(desc->max_batch) * ob.bytes_per_example() / 16.0;
    const double pixels = static_cast<double>(desc->max_batch) * op.oh * op.ow;
    if (pieces >= 2147483648.0 || pixels >= 67108864.0) {
      return dv::fail(DV_ERR_INVALID_ARGUMENT,
                      "dv_model_create: max_batch too large for this image size (N*OH*OW must stay "
                      "below 2^26 and every activation tensor below 2^31 16-byte pieces)");
    }
  }
  if (static_cast<double>(desc->max_batch) * desc->height * desc->width * desc->channels >=
      2147483648.0) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_model_create: max_batch * H * W * C must be < 2^31");
  }
  m->dbuf.resize(m->buffers.size());
  for (size_t i = 0; i < m->buffers.size(); ++i) {
    const BufferDesc& b = m->buffers[i];
    const bool stem_buf = static_cast<int>(i) < m->stem_out_buf;
    // imgconv tiles read whole groups of images and (1x1 steps of 4 channel chunks) up to
    // three chunks past the last channel: both stay inside the allocation, which is zeroed
    // to its full capacity (finite values times zero-padded weights).
    const size_t examples = std::max(
        static_cast<size_t>(stem_buf ? std::min(desc->max_batch, stem_sub_batch()) : desc->max_batch),
        static_cast<size_t>(b.min_examples));
    const size_t bytes = examples * b.bytes_per_example() +
                         static_cast<size_t>(8) * (b.h + 2 * b.halo) * (b.w + 2 * b.halo) * 16;
    if (int rc = m->dbuf[i].reserve(bytes)) return rc;
    DV_HIP_CHECK(hipMemset(m->dbuf[i].ptr, 0, m->dbuf[i].cap));  // halos stay zero forever
  }
  if (int rc = m->d_w.reserve(m->packed_halfs * 2)) return rc;
  if (int rc = m->d_shift.reserve(m->shift_floats * 4)) return rc;
  if (int rc = m->d_dense_w.reserve(static_cast<size_t>(m->feat_c) * desc->num_classes * 4)) return rc;
  if (int rc = m->d_dense_b.reserve(desc->num_classes * 4)) return rc;
  if (int rc = m->d_ext.reserve(sizeof(ExtPtrs))) return rc;
  // Skip the stem work that only sees the zero rows below the pile-up (on unless DV_BLANK_SKIP=0): needs the uint8
  // front end (the scan reads the caller's image), a single-branch 3x3 80->192 and whole dwords per image
  if (!(getenv("DV_BLANK_SKIP") != nullptr && atoi(getenv("DV_BLANK_SKIP")) == 0) &&
      m->ops[0].first_u8 && m->blank_conv4_op >= 0 && m->ops[m->blank_conv4_op].group_followers == 0 &&
      (static_cast<size_t>(desc->height) * desc->width * desc->channels) % 4 == 0) {
    if (int rc = m->d_blank_thr.reserve(static_cast<size_t>(7) * desc->max_batch * sizeof(int))) return rc;
    DV_HIP_CHECK(hipMemset(m->d_blank_thr.ptr, 0, m->d_blank_thr.cap));
    m->blank_skip = true;
  }
  *out = m.release();
  return DV_OK;
}

void dv_model_destroy(dv_model* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  for (auto& b : m->dbuf) b.release();
  m->d_w.release();
  m->d_shift.release();
  m->d_dense_w.release();
  m->d_dense_b.release();
  m->d_tbl.release();
  m->d_blank_thr.release();
  m->d_ext.release();
  m->d_blank_conv4.release();
  m->d_blank_c2.release();
  m->d_blank_b.release();
  for (auto& g : m->graphs) (void)hipGraphExecDestroy(g.exec);
  delete m;
}

int64_t dv_model_num_params(const dv_model* m) { return m ? m->n_params : 0; }

int64_t dv_model_conv_macs(const dv_model* m) {
  if (!m) return 0;
  int64_t macs = 0;
  for (const Op& op : m->ops) {
    if (op.type == kOpConv) {
      macs += static_cast<int64_t>(op.kh) * op.kw * op.cin_real * op.cout * op.oh * op.ow;
    }
  }
  return macs;
}

int dv_model_num_layers(const dv_model* m) {
  return m ? static_cast<int>(m->layers.size()) : 0;
}

int dv_model_layer_info(const dv_model* m, int layer, int32_t* kh, int32_t* kw,
                        int32_t* cin, int32_t* cout, int64_t* param_offset) {
  if (!m || layer < 0 || layer >= static_cast<int>(m->layers.size())) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_model_layer_info: bad layer");
  }
  const LayerInfo& l = m->layers[layer];
  if (kh) *kh = l.kh;
  if (kw) *kw = l.kw;
  if (cin) *cin = l.cin;
  if (cout) *cout = l.cout;
  if (param_offset) *param_offset = l.param_off;
  return DV_OK;
}

static int enqueue_forward(dv_model* m, int n, hipStream_t stream);
static void set_ext(dv_model* m, const uint8_t* images, float* probs, hipStream_t stream,
                    const int32_t* rows_hint = nullptr, int rows_add = 0) {
  hipLaunchKernelGGL(set_ext_kernel, dim3(1), dim3(1), 0, stream, static_cast<ExtPtrs*>(m->d_ext.ptr), images, probs,
                     rows_hint, rows_add);
}

// Blank-row skipping: the stem's response to the all-blank (all-zero) image, computed once per
// set of weights by the ordinary kernels -- the same arithmetic that produces those values
// inside a real image -- and kept as the source of the blank tiles.
static int prepare_blank_responses(dv_model* m) {
  m->blank_ready = false;
  if (!m->blank_skip) return DV_OK;
  for (auto& g : m->graphs) {   // captured without the blank arguments
    (void)hipStreamSynchronize(g.stream);
    (void)hipGraphExecDestroy(g.exec);
  }
  m->graphs.clear();
  const size_t img_bytes = static_cast<size_t>(m->desc.height) * m->desc.width * m->desc.channels;
  dv::DeviceBuffer zero_img, probs;
  int rc = zero_img.reserve(img_bytes);
  if (rc == DV_OK) rc = probs.reserve(sizeof(float) * m->desc.num_classes);
  if (rc == DV_OK && hipMemset(zero_img.ptr, 0, img_bytes) != hipSuccess) rc = dv::fail(DV_ERR_HIP, "hipMemset");
  if (rc == DV_OK) {
    set_ext(m, static_cast<const uint8_t*>(zero_img.ptr), static_cast<float*>(probs.ptr), nullptr);
    rc = enqueue_forward(m, 1, nullptr);
  }
  if (rc == DV_OK && hipDeviceSynchronize() != hipSuccess) rc = dv::fail(DV_ERR_HIP, "blank forward failed");
  auto keep = [&](int buf, dv::DeviceBuffer* dst) {
    if (rc != DV_OK || buf < 0 || m->buffers[buf].h <= 1) return;   // (LDS-only tensors have no buffer)
    const size_t bytes = m->buffers[buf].bytes_per_example();
    rc = dst->reserve(bytes);
    if (rc == DV_OK && hipMemcpy(dst->ptr, m->dbuf[buf].ptr, bytes, hipMemcpyDeviceToDevice) != hipSuccess) {
      rc = dv::fail(DV_ERR_HIP, "copying the blank response");
    }
  };
  keep(m->ops[m->blank_conv4_op].out_buf, &m->d_blank_conv4);
  keep(m->ops[1].out_buf, &m->d_blank_c2);                               // conv2 (stem_a's output)
  if (m->ops[2].stem_b) keep(m->ops[3].out_buf, &m->d_blank_b);          // the 1x1 64->80 (stem_b's output)
  zero_img.release();
  probs.release();
  if (rc != DV_OK) return rc;
  m->blank_ready = true;
  return DV_OK;
}

int dv_model_load_weights(dv_model* m, const float* weights, int64_t n) {
  if (!m || !weights) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_model_load_weights: null");
  if (n != m->n_params) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT,
                    "dv_model_load_weights: expected " + std::to_string(m->n_params) +
                        " values, got " + std::to_string(n));
  }
  DV_HIP_CHECK(hipSetDevice(m->device));
  std::vector<_Float16> packed(m->packed_halfs, static_cast<_Float16>(0.f));
  std::vector<float> shift(m->shift_floats, 0.f);
  for (const Op& op : m->ops) {
    if (op.type != kOpConv) continue;
    const LayerInfo& l = m->layers[op.layer];
    const float* var = weights + l.param_off + static_cast<size_t>(l.kh) * l.kw * l.cin * l.cout + 2 * l.cout;
    for (int co = 0; co < l.cout; ++co) {
      if (!(var[co] + 1e-3f > 0.f)) return dv::fail(DV_ERR_BAD_INPUT, "non-positive BatchNorm variance");
    }
  }
  // One conv layer's weights -> its fragment image.  Layers write disjoint parts of `packed` and
  // `shift` (the siblings of a grouped launch share a region but own different cout rows), so
  // they are packed on a few host threads: the 22 M weights are read with a stride of cout and
  // one thread needs 0.2-0.3 s for them -- as long as a 100 kb make_examples run takes.
  auto pack_op = [&](size_t oi) {
    const Op& op = m->ops[oi];
    if (op.type != kOpConv) return;
    const LayerInfo& l = m->layers[op.layer];
    const float* w = weights + l.param_off;  // HWIO
    const size_t wn = static_cast<size_t>(l.kh) * l.kw * l.cin * l.cout;
    const float* beta = w + wn;
    const float* mean = beta + l.cout;
    const float* var = mean + l.cout;
    std::vector<float> inv(l.cout);
    for (int co = 0; co < l.cout; ++co) {
      inv[co] = 1.0f / std::sqrt(var[co] + 1e-3f);
      shift[op.shift_off + co] = beta[co] - mean[co] * inv[co];
    }
    if ((m->ops[0].stem_a && oi < 2) || (m->ops[2].stem_b && (oi == 2 || oi == 3))) {
      _Float16* dst = packed.data() + op.w_off;   // fused stem: stem.hip's own fragment images
      if (oi == 0 && l.cin > 8) dv::pack_stem_a_w1_wide(w, inv.data(), l.cin, dst);
      if (oi == 0 && l.cin <= 8) dv::pack_stem_a_w1(w, inv.data(), l.cin, dst);
      if (oi == 1) dv::pack_stem_a_w2(w, inv.data(), dst);
      if (oi == 2) dv::pack_stem_b_w3(w, inv.data(), dst);
      if (oi == 3) dv::pack_stem_b_w4(w, inv.data(), l.cout, dst);
      return;
    }
    if (op.first_u8) {
      // C <= 8:  [chunk kc][k-group g = tap 2kc+g][cout][8]: channel c < cin_real, else zero
      // C <= 16: [chunk kc = tap][k-group g = channels 8g..8g+7][cout][8]
      const bool wide = l.cin > 8;
      for (int kc = 0; kc < op.n_chunks; ++kc)
        for (int g = 0; g < 2; ++g) {
          const int tap = wide ? kc : 2 * kc + g;
          if (tap >= op.kh * op.kw) continue;
          const int kh = tap / op.kw, kw = tap % op.kw;
          for (int co = 0; co < op.cout; ++co)
            for (int ci = wide ? 8 * g : 0; ci < (wide ? std::min(l.cin, 8 * g + 8) : l.cin); ++ci) {
              const float v = w[((static_cast<size_t>(kh) * l.kw + kw) * l.cin + ci) * l.cout + co];
              packed[op.w_off + ((static_cast<size_t>(kc) * 2 + g) * 32 + co) * 8 + (ci & 7)] =
                  static_cast<_Float16>(v * inv[co]);
            }
        }
      return;
    }
    if (op.chain_len > 0 || op.in_chain) {
      // chain.hip: [channel chunk][tap][k-group][cout_pad][8]
      const int cout_pad = (op.cout + 31) / 32 * 32, taps = op.kh * op.kw;
      _Float16* dst = packed.data() + op.w_off;
      for (int cc = 0; cc < l.cin / kChunk; ++cc)
        for (int tap = 0; tap < taps; ++tap) {
          const int kh = tap / op.kw, kw = tap % op.kw;
          for (int co = 0; co < op.cout; ++co)
            for (int jj = 0; jj < kChunk; ++jj) {
              const int ci = cc * kChunk + jj;
              const float v = w[((static_cast<size_t>(kh) * l.kw + kw) * l.cin + ci) * l.cout + co];
              dst[(((static_cast<size_t>(cc) * taps + tap) * 2 + jj / 8) * cout_pad + co) * 8 + (jj % 8)] =
                  static_cast<_Float16>(v * inv[co]);
            }
        }
      return;
    }
    // Row of this op's cout `co` in the launch's concatenated cout space: the siblings
    // grouped before it (leader first) each occupy whole 32-cout subtiles.
    int sub0 = 0;
    {
      size_t lead = oi;
      while (lead > 0 && m->ops[lead].group_followers == 0 && m->ops[lead - 1].type == kOpConv &&
             m->ops[lead - 1].w_off == op.w_off) {
        --lead;  // walk back to the leader (all ops of a launch share w_off)
      }
      for (size_t j = lead; j < oi; ++j) sub0 += (m->ops[j].cout + 31) / 32;
    }
    const int bn = op.nb * 32;
    const int taps = op.kh * op.kw;
    {
      // the launch's leader carries the imgconv decision
      size_t lead = oi;
      while (lead > 0 && m->ops[lead].group_followers == 0 && m->ops[lead - 1].type == kOpConv &&
             m->ops[lead - 1].w_off == op.w_off) {
        --lead;
      }
      const Op& lo = m->ops[lead];
      if (lo.v2) {
        // [cout tile][step][chunk in step][tap][k-group][bn couts][8]
        const int kcs = dv::imgconv_kc(op.kh, op.kw);
        const size_t slab = dv::imgconv_wslab_halfs(op.kh, op.kw, op.nb);
        for (int cc = 0; cc < l.cin / kChunk + (l.cin % kChunk ? 1 : 0); ++cc) {
          const int st = cc / kcs, kc = cc % kcs;
          for (int tap = 0; tap < taps; ++tap) {
            const int kh = tap / op.kw, kw = tap % op.kw;
            for (int co = 0; co < op.cout; ++co) {
              const int row = sub0 * 32 + co;
              const int t = row / bn, r = row % bn;
              _Float16* sub = packed.data() + op.w_off +
                              (static_cast<size_t>(t) * lo.v2_steps + st) * slab +
                              (static_cast<size_t>(kc) * taps + tap) * 2 * bn * 8;
              for (int jj = 0; jj < kChunk; ++jj) {
                const int ci = cc * kChunk + jj;
                if (ci >= l.cin) continue;
                const float v = w[((static_cast<size_t>(kh) * l.kw + kw) * l.cin + ci) * l.cout + co];
                sub[(static_cast<size_t>(jj / 8) * bn + r) * 8 + (jj % 8)] = static_cast<_Float16>(v * inv[co]);
              }
            }
          }
        }
        return;
      }
    }
    // row-band mode: one image per output row `band_r`, holding the op.band tap rows
    // kh = pad_h - band_r + 0..band-1 that meet map rows 0..band-1
    const int eff_taps = op.band ? op.band * op.kw : taps;
    const int n_tiles_op = ((op.cout + 31) / 32 + op.nb - 1) / op.nb;   // band ops are never grouped
    // split rows: chunk 2q = W_hi, chunk 2q + 1 = W_lo of pixel chunk q; plain rows of a split launch
    // (siblings that are not split) use the first half of the launch's chunk slots
    const int parts = op.split_rows ? 2 : 1;
    const int op_chunks = op.split && !op.split_rows ? op.n_chunks / 2 : op.n_chunks;
    for (int band_r = 0; band_r < (op.band ? op.band : 1); ++band_r)
    for (int kc = 0; kc < op_chunks; ++kc) {
      const int sl = kc / kSlabChunks, j = kc % kSlabChunks;
      const int q = kc / parts, part = kc % parts;
      const int cc = q / eff_taps, tap = q % eff_taps;  // chunk-major, tap-minor (ChunkWalk)
      const int kh = tap / op.kw + (op.band ? op.pad_h - band_r : 0), kw = tap % op.kw;
      for (int co = 0; co < op.cout; ++co) {
        const int row = sub0 * 32 + co;
        const int t = row / bn + band_r * n_tiles_op, r = row % bn;
        // chunk image [k-group g][cout r][8]: matches conv_mfma_kernel's frag_off
        _Float16* chunk = packed.data() + op.w_off +
                          ((static_cast<size_t>(t) * op.n_steps + sl) * kSlabChunks + j) * bn * kChunk;
        for (int jj = 0; jj < kChunk; ++jj) {
          const int ci = cc * kChunk + jj;
          if (ci >= l.cin) continue;  // padded input channels
          const float v = w[((static_cast<size_t>(kh) * l.kw + kw) * l.cin + ci) * l.cout + co] * inv[co];
          const _Float16 hi = static_cast<_Float16>(v);
          chunk[(static_cast<size_t>(jj / 8) * bn + r) * 8 + (jj % 8)] =
              part == 0 ? hi : static_cast<_Float16>(v - static_cast<float>(hi));
        }
      }
    }
  };
  {
    std::atomic<size_t> next{0};
    auto work = [&] {
      for (size_t oi = next.fetch_add(1); oi < m->ops.size(); oi = next.fetch_add(1)) pack_op(oi);
    };
    const unsigned helpers = std::min(7u, std::max(1u, std::thread::hardware_concurrency()) - 1);
    std::vector<std::thread> threads;
    for (unsigned t = 0; t < helpers; ++t) threads.emplace_back(work);
    work();
    for (std::thread& t : threads) t.join();
  }
  const LayerInfo& dl = m->layers.back();
  const float* dw = weights + dl.param_off;
  DV_HIP_CHECK(hipMemcpy(m->d_w.ptr, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
  DV_HIP_CHECK(hipMemcpy(m->d_shift.ptr, shift.data(), shift.size() * 4, hipMemcpyHostToDevice));
  DV_HIP_CHECK(hipMemcpy(m->d_dense_w.ptr, dw, static_cast<size_t>(dl.cin) * dl.cout * 4,
                         hipMemcpyHostToDevice));
  DV_HIP_CHECK(hipMemcpy(m->d_dense_b.ptr, dw + static_cast<size_t>(dl.cin) * dl.cout,
                         dl.cout * 4, hipMemcpyHostToDevice));
  m->h_shift = shift;
  m->h_dense_b.assign(dw + static_cast<size_t>(dl.cin) * dl.cout, dw + static_cast<size_t>(dl.cin) * dl.cout + dl.cout);
  m->loaded = true;
  return prepare_blank_responses(m);
}

// The op list as calib.h's plan of plain NHWC tensors: fused pools unfolded (pool_in / pool_out), LDS-only
// tensors given their real size, and the tensors the product keeps wider than fp16 marked (keep_f32).
static dv::CalibPlan calib_plan_of(const dv_model* m) {
  dv::CalibPlan plan;
  plan.bufs.resize(m->buffers.size());
  for (size_t b = 0; b < m->buffers.size(); ++b) plan.bufs[b] = {m->buffers[b].h, m->buffers[b].w, m->buffers[b].c};
  plan.bufs[0] = {m->desc.height, m->desc.width, m->desc.channels};
  for (const Op& op : m->ops) {
    dv::CalibOp c{};
    c.type = op.type == kOpConv ? 0 : op.type == kOpMaxPool ? 1 : 2;
    c.in_buf = op.in_buf;
    c.out_buf = op.out_buf;
    c.out_coff = op.out_coff;
    c.shift_off = -1;
    int oh = op.oh, ow = op.ow;
    if (op.type == kOpConv) {
      c.pool_in = op.pool_in;
      c.pool_out = op.pool_out;
      c.kh = op.kh;
      c.kw = op.kw;
      c.stride = op.stride;
      c.pad_h = op.pad_h;
      c.pad_w = op.pad_w;
      c.cin = op.cin_real;
      c.cout = op.cout;
      c.w_off = m->layers[op.layer].param_off;
      c.raw = op.raw;
      c.shift_off = static_cast<int64_t>(op.shift_off);
      c.split = op.split_rows;
      c.keep_f32 = m->buffers[op.out_buf].f32 || m->buffers[op.out_buf].wide ? 1 : 0;
      if (op.pool_in) {   // op.ih / op.iw: the tensor as stored, before the on-the-fly pool
        plan.bufs[op.in_buf].h = op.ih;
        plan.bufs[op.in_buf].w = op.iw;
      }
      if (op.pool_out) {
        oh = (oh - 3) / 2 + 1;
        ow = (ow - 3) / 2 + 1;
      }
    } else if (op.type == kOpAvgPool) {
      c.shift_relu = op.pool_shift_relu;
      if (op.pool_shift_relu) c.shift_off = static_cast<int64_t>(op.shift_off);
      c.cout = op.cout;
      c.keep_f32 = m->buffers[op.out_buf].f32 || m->buffers[op.out_buf].wide ? 1 : 0;
    } else {
      c.cout = op.cout;
    }
    plan.bufs[op.out_buf].h = oh;   // LDS-only tensors of the fused kernels are 1 x 1 in `buffers`
    plan.bufs[op.out_buf].w = ow;
    plan.ops.push_back(c);
  }
  plan.feat_buf = m->feat_buf;
  plan.num_classes = m->desc.num_classes;
  plan.dense_off = m->layers.back().param_off;
  return plan;
}

// Shift calibration (include/dvhip.h, csrc/calib.h): the op list as a plan of plain NHWC tensors --
// fused pools unfolded (pool_in / pool_out), LDS-only tensors given their real size -- run through
// the two fp32 pipelines; shifts and the Dense bias move by the mean differences.
int dv_model_calibrate(dv_model* m, const float* weights, int64_t n_weights, const uint8_t* images, int n_images,
                       float* corrections, int64_t capacity) {
  if (!m || !weights || !images) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_model_calibrate: null");
  if (!m->loaded) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_model_calibrate: load weights first");
  if (n_weights != m->n_params) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_model_calibrate: wrong number of weights");
  if (n_images < 1 || n_images > 4096) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_model_calibrate: 1..4096 images");
  const dv::CalibPlan plan = calib_plan_of(m);
  std::vector<float> corr, dense_corr;
  if (int rc = dv::run_calibration(plan, m->device, weights, n_weights, m->h_shift, images, n_images, &corr,
                                   &dense_corr)) {
    return rc;
  }
  std::vector<float> shift = m->h_shift, dense_b = m->h_dense_b;
  for (size_t i = 0; i < shift.size(); ++i) shift[i] -= corr[i];
  for (size_t k = 0; k < dense_b.size(); ++k) dense_b[k] -= dense_corr[k];
  DV_HIP_CHECK(hipSetDevice(m->device));
  DV_HIP_CHECK(hipDeviceSynchronize());
  DV_HIP_CHECK(hipMemcpy(m->d_shift.ptr, shift.data(), shift.size() * 4, hipMemcpyHostToDevice));
  DV_HIP_CHECK(hipMemcpy(m->d_dense_b.ptr, dense_b.data(), dense_b.size() * 4, hipMemcpyHostToDevice));
  if (corrections != nullptr) {
    int64_t at = 0;
    std::vector<const Op*> by_layer(m->layers.size(), nullptr);
    for (const Op& op : m->ops) {
      if (op.type == kOpConv) by_layer[op.layer] = &op;
    }
    for (size_t l = 0; l + 1 < m->layers.size(); ++l) {
      for (int co = 0; co < by_layer[l]->cout && at < capacity; ++co) corrections[at++] = corr[by_layer[l]->shift_off + co];
    }
    for (size_t k = 0; k < dense_corr.size() && at < capacity; ++k) corrections[at++] = dense_corr[k];
  }
  return prepare_blank_responses(m);
}

int dv_model_apply_corrections(dv_model* m, const float* corrections, int64_t n) {
  if (!m || !corrections) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_model_apply_corrections: null");
  if (!m->loaded) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_model_apply_corrections: load weights first");
  std::vector<const Op*> by_layer(m->layers.size(), nullptr);
  for (const Op& op : m->ops) {
    if (op.type == kOpConv) by_layer[op.layer] = &op;
  }
  int64_t want = static_cast<int64_t>(m->h_dense_b.size());
  for (size_t l = 0; l + 1 < m->layers.size(); ++l) want += by_layer[l]->cout;
  if (n != want) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_model_apply_corrections: wrong number of corrections");
  for (int64_t i = 0; i < n; ++i) {
    if (!std::isfinite(corrections[i])) return dv::fail(DV_ERR_BAD_INPUT, "dv_model_apply_corrections: non-finite correction");
  }
  std::vector<float> shift = m->h_shift, dense_b = m->h_dense_b;
  int64_t at = 0;
  for (size_t l = 0; l + 1 < m->layers.size(); ++l) {
    for (int co = 0; co < by_layer[l]->cout; ++co) shift[by_layer[l]->shift_off + co] -= corrections[at++];
  }
  for (size_t k = 0; k < dense_b.size(); ++k) dense_b[k] -= corrections[at++];
  DV_HIP_CHECK(hipSetDevice(m->device));
  DV_HIP_CHECK(hipDeviceSynchronize());
  DV_HIP_CHECK(hipMemcpy(m->d_shift.ptr, shift.data(), shift.size() * 4, hipMemcpyHostToDevice));
  DV_HIP_CHECK(hipMemcpy(m->d_dense_b.ptr, dense_b.data(), dense_b.size() * 4, hipMemcpyHostToDevice));
  return prepare_blank_responses(m);
}

// Diagnostic (include/dvhip.h): the calibration's two fp32 pipelines as a probe of WHERE the fp16 error enters.
int dv_model_num_ops(const dv_model* m) { return m ? static_cast<int>(m->ops.size()) : 0; }

int dv_model_op_label(const dv_model* m, int op_index, char* buf, int capacity) {
  if (!m || !buf || capacity < 1 || op_index < 0 || op_index >= static_cast<int>(m->ops.size())) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_model_op_label: bad argument");
  }
  const Op& op = m->ops[op_index];
  const char* kind = op.type == kOpConv ? "conv" : op.type == kOpMaxPool ? "maxpool" : "avgpool";
  snprintf(buf, static_cast<size_t>(capacity), "%s layer=%d k=%dx%d s=%d cin=%d cout=%d out=%dx%d raw=%d in_buf=%d out_buf=%d coff=%d lds_only=%d",
           kind, op.layer, op.kh, op.kw, op.stride, op.cin, op.cout, op.oh, op.ow, op.raw ? 1 : 0, op.in_buf, op.out_buf,
           op.out_coff, (op.type == kOpConv && (op.stem_a || op.stem_b)) || (op_index + 1 < static_cast<int>(m->ops.size()) && m->ops[op_index + 1].in_chain && m->ops[op_index + 1].in_buf == op.out_buf) ? 1 : 0);
  return DV_OK;
}

int dv_model_probe_rounding(dv_model* m, const float* weights, int64_t n_weights, const uint8_t* images, int n_images,
                            const uint8_t* keep_f32, int flags, const float* corrections, int64_t n_corrections,
                            float* logits_r, float* logits_e, float* corrections_out) {
  if (!m || !weights || !images || !logits_e) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_model_probe_rounding: null");
  if (!m->loaded) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_model_probe_rounding: load weights first");
  if (n_weights != m->n_params) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_model_probe_rounding: wrong number of weights");
  if (n_images < 1 || n_images > 4096) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_model_probe_rounding: 1..4096 images");
  dv::CalibPlan plan = calib_plan_of(m);
  if (keep_f32 != nullptr) {
    for (size_t i = 0; i < plan.ops.size(); ++i) plan.ops[i].keep_f32 = keep_f32[i] ? 1 : 0;
  }
  std::vector<float> corr_in(m->h_shift.size(), 0.f), dense_in(m->h_dense_b.size(), 0.f);
  if (corrections != nullptr) {
    std::vector<const Op*> by_layer(m->layers.size(), nullptr);
    for (const Op& op : m->ops) {
      if (op.type == kOpConv) by_layer[op.layer] = &op;
    }
    int64_t want = static_cast<int64_t>(dense_in.size());
    for (size_t l = 0; l + 1 < m->layers.size(); ++l) want += by_layer[l]->cout;
    if (n_corrections != want) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_model_probe_rounding: wrong number of corrections");
    int64_t at = 0;
    for (size_t l = 0; l + 1 < m->layers.size(); ++l) {
      for (int co = 0; co < by_layer[l]->cout; ++co) corr_in[by_layer[l]->shift_off + co] = corrections[at++];
    }
    for (size_t k = 0; k < dense_in.size(); ++k) dense_in[k] = corrections[at++];
  }
  const bool measure = (flags & 2) != 0;   // the calibration proper under this plan: corrections measured on these images
  if (measure && corrections != nullptr) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_model_probe_rounding: measure or apply");
  dv::CalibProbe probe;
  probe.corr_in = measure ? nullptr : corr_in.data();
  probe.dense_corr_in = measure ? nullptr : dense_in.data();
  probe.logits_r = logits_r;
  probe.logits_e = logits_e;
  probe.skip_r = logits_r == nullptr && !measure;
  probe.weights_f32 = (flags & 1) != 0;
  std::vector<float> corr, dense_corr;
  if (int rc = dv::run_calibration(plan, m->device, weights, n_weights, m->h_shift, images, n_images, &corr, &dense_corr,
                                   &probe)) {
    return rc;
  }
  if (corrections_out != nullptr) {
    std::vector<const Op*> by_layer(m->layers.size(), nullptr);
    for (const Op& op : m->ops) {
      if (op.type == kOpConv) by_layer[op.layer] = &op;
    }
    int64_t at = 0;
    for (size_t l = 0; l + 1 < m->layers.size(); ++l) {
      for (int co = 0; co < by_layer[l]->cout; ++co) corrections_out[at++] = corr[by_layer[l]->shift_off + co];
    }
    for (float v : dense_corr) corrections_out[at++] = v;
  }
  return DV_OK;
}

// Testing hook: copies activation buffer `index` (NHWC fp16, first n examples)
// to host memory; returns its shape.  Buffer 0 is the preprocessed input.
int dv_model_debug_tensor(dv_model* m, int index, int n, void* host_out, int32_t* h,
                          int32_t* w, int32_t* c) {
  if (m && index == -1) index = m->feat_buf;      // the last block's output (input of the head)
  if (m && index == -2) index = m->stem_out_buf;  // the stem's output (input of mixed0)
  if (!m || index < 0 || index >= static_cast<int>(m->buffers.size())) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_model_debug_tensor: bad index");
  }
  const BufferDesc& b = m->buffers[index];
  // reports the PADDED plane size (interior + 2*halo on each axis)
  if (h) *h = b.h + 2 * b.halo;
  if (w) *w = b.w + 2 * b.halo;
  if (c) *c = b.c;
  if (host_out) {
    DV_HIP_CHECK(hipSetDevice(m->device));
    DV_HIP_CHECK(hipDeviceSynchronize());
    if (b.f32) {   // float32 tensors (BufferDesc::f32) leave as the fp16 numbers the hook's contract promises
      std::vector<float> tmp(static_cast<size_t>(n) * b.bytes_per_example() / 4);
      DV_HIP_CHECK(hipMemcpy(tmp.data(), m->dbuf[index].ptr, tmp.size() * 4, hipMemcpyDeviceToHost));
      _Float16* dst = static_cast<_Float16*>(host_out);
      for (size_t i = 0; i < tmp.size(); ++i) dst[i] = static_cast<_Float16>(tmp[i]);
    } else if (b.wide) {   // wide tensors (precise mode) leave as hi + lo, rounded to the fp16 the hook's contract promises
      const size_t half_halfs = b.bytes_per_example() / 4;   // fp16 numbers of the hi (= of the lo) part of one example
      std::vector<_Float16> tmp(static_cast<size_t>(n) * half_halfs * 2);
      DV_HIP_CHECK(hipMemcpy(tmp.data(), m->dbuf[index].ptr, tmp.size() * 2, hipMemcpyDeviceToHost));
      _Float16* dst = static_cast<_Float16*>(host_out);
      for (int e = 0; e < n; ++e)
        for (size_t i = 0; i < half_halfs; ++i) {
          dst[static_cast<size_t>(e) * half_halfs + i] = static_cast<_Float16>(
              static_cast<float>(tmp[static_cast<size_t>(e) * 2 * half_halfs + i]) +
              static_cast<float>(tmp[static_cast<size_t>(e) * 2 * half_halfs + half_halfs + i]));
        }
    } else {
      DV_HIP_CHECK(hipMemcpy(host_out, m->dbuf[index].ptr,
                             static_cast<size_t>(n) * b.bytes_per_example(),
                             hipMemcpyDeviceToHost));
    }
  }
  return DV_OK;
}

// Enqueues the whole forward for `n` examples on `stream` (eager launches).  The caller's image
// and probability pointers come from the device-side table (set_ext), not from kernel arguments.
static int enqueue_forward(dv_model* m, int n, hipStream_t stream) {
  const size_t img_bytes = static_cast<size_t>(m->desc.height) * m->desc.width * m->desc.channels;
  const ExtPtrs* ext = static_cast<const ExtPtrs*>(m->d_ext.ptr);
  // split evenly so that no launch is left with a sliver of a batch
  const int n_parts = (n + m->desc.max_batch - 1) / m->desc.max_batch;
  const int part = n_parts ? (n + n_parts - 1) / n_parts : 0;
  for (int done = 0; done < n; done += part) {
    const int nb = std::min(part, n - done);
    for (int sb0 = 0; sb0 < nb; sb0 += stem_sub_batch()) {
      const int sb = std::min(stem_sub_batch(), nb - sb0);
      const size_t img_off = static_cast<size_t>(done + sb0) * img_bytes;
      if (m->blank_on()) {
        dv::ProfileScope prof(dv::kProfOther, stream);
        hipLaunchKernelGGL(blank_rows_kernel, dim3(sb), dim3(256), 0, stream, ext, img_off, done + sb0, m->desc.height,
                           m->desc.width * m->desc.channels, static_cast<int*>(m->d_blank_thr.ptr),
                           m->desc.max_batch);
        const Op& c4 = m->ops[m->blank_conv4_op];
        hipLaunchKernelGGL(blank_need_kernel, dim3((sb + 255) / 256), dim3(256), 0, stream,
                           static_cast<int*>(m->d_blank_thr.ptr), m->desc.max_batch, sb, m->ops[1].oh, m->ops[3].oh,
                           c4.ow, c4.pool_out ? m->buffers[c4.out_buf].h : c4.oh, m->ops[2].stem_b ? 1 : 0,
                           c4.pool_out ? 1 : 0);
      }
      if (!m->ops[0].first_u8) {
        const size_t n_pix = static_cast<size_t>(sb) * m->desc.height * m->desc.width;
        dv::ProfileScope prof(dv::kProfOther, stream);
        hipLaunchKernelGGL(preprocess_kernel, dim3(static_cast<unsigned>((n_pix + 255) / 256)),
                           dim3(256), 0, stream, ext, img_off, static_cast<_Float16*>(m->dbuf[0].ptr),
                           n_pix, m->desc.channels, m->desc.height, m->desc.width,
                           m->buffers[0].geom());
      }
      if (int rc = run_ops(m, 0, m->stem_ops_end, sb, stream, m->stem_out_buf, sb0, img_off)) return rc;
    }
    if (int rc = run_ops(m, m->stem_ops_end, static_cast<int>(m->ops.size()), nb, stream)) {
      return rc;
    }
    {
      dv::ProfileScope prof(dv::kProfOther, stream);
      hipLaunchKernelGGL(head_kernel, dim3(nb), dim3(256), 0, stream,
                         static_cast<const float*>(m->dbuf[m->feat_buf].ptr),
                         static_cast<const float*>(m->d_dense_w.ptr),
                         static_cast<const float*>(m->d_dense_b.ptr), ext,
                         static_cast<size_t>(done) * m->desc.num_classes,
                         m->buffers[m->feat_buf].geom(), m->desc.num_classes);
    }
  }
  return DV_OK;
}

int dv_model_infer(dv_model* m, const uint8_t* images, int n, float* probs, void* stream_v) {
  return dv_model_infer_rows(m, images, n, probs, nullptr, 0, stream_v);
}

int dv_model_infer_rows(dv_model* m, const uint8_t* images, int n, float* probs, const int32_t* rows_used, int rows_add,
                        void* stream_v) {
  if (!m || !images || !probs || n < 0) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_model_infer: bad argument");
  }
  if (!m->loaded) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_model_infer: no weights loaded");
  if (n == 0) return DV_OK;
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  DV_HIP_CHECK(hipSetDevice(m->device));
  // The forward is ~65 launches; replaying it as one hipGraph removes the per-launch gaps
  // (measured +0.6 % at 8 K examples per forward, more for small batches).  Graphs are keyed by
  // (n, stream) only: the caller's pointers travel through the ExtPtrs table, written in stream
  // order ahead of every forward, so fresh image tensors per region replay the same graph.
  // Per-launch event profiling and DV_OP_TRACE need eager launches, and the legacy default
  // stream cannot be captured.
  static const bool op_trace = getenv("DV_OP_TRACE") != nullptr;
  static const bool no_graph = getenv("DV_NO_GRAPH") != nullptr || op_trace;
  // a caller that is already capturing this stream gets plain launches (they join ITS graph)
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  if (stream != nullptr) (void)hipStreamIsCapturing(stream, &capturing);
  set_ext(m, images, probs, stream, rows_used, rows_add);
  if (no_graph || dv::profiling_enabled() || stream == nullptr ||
      capturing != hipStreamCaptureStatusNone) {
    static std::vector<OpTrace> trace_store;
    g_trace = op_trace ? &trace_store : nullptr;
    if (int rc = enqueue_forward(m, n, stream)) return rc;
    DV_HIP_CHECK(hipGetLastError());
    if (op_trace) dump_trace(stream);
    return DV_OK;
  }
  for (const dv_model::GraphEntry& g : m->graphs) {
    if (g.n == n && g.stream == stream) {
      DV_HIP_CHECK(hipGraphLaunch(g.exec, stream));
      ++m->graph_replays;
      return DV_OK;
    }
  }
  hipGraph_t graph = nullptr;
  DV_HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed));
  const int rc = enqueue_forward(m, n, stream);
  const hipError_t ce = hipStreamEndCapture(stream, &graph);
  if (rc != DV_OK) {
    if (graph) (void)hipGraphDestroy(graph);
    return rc;
  }
  if (ce != hipSuccess || graph == nullptr) {
    return dv::fail(DV_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
  }
  dv_model::GraphEntry e{n, stream, nullptr};
  const hipError_t ie = hipGraphInstantiate(&e.exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (ie != hipSuccess) {
    return dv::fail(DV_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ie));
  }
  if (m->graphs.size() >= 8) {  // bounded cache; the evicted replay may still be in flight
    (void)hipStreamSynchronize(m->graphs.front().stream);
    (void)hipGraphExecDestroy(m->graphs.front().exec);
    m->graphs.erase(m->graphs.begin());
  }
  m->graphs.push_back(e);
  ++m->graph_captures;
  DV_HIP_CHECK(hipGraphLaunch(e.exec, stream));
  return DV_OK;
}

// 1 when the model runs in precise mode (include/dvhip.h).
int dv_model_is_precise(const dv_model* m) { return m && m->precise ? 1 : 0; }

// Blank-row skipping on / off at run time (include/dvhip.h): bench.py times the dense path on the same model.
int dv_model_set_blank_skip(dv_model* m, int enabled) {
  if (!m) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_model_set_blank_skip: null");
  const bool on = enabled != 0;
  if (on == m->blank_enabled) return DV_OK;
  DV_HIP_CHECK(hipSetDevice(m->device));
  for (auto& g : m->graphs) {   // captured with the other setting
    (void)hipStreamSynchronize(g.stream);
    (void)hipGraphExecDestroy(g.exec);
  }
  m->graphs.clear();
  m->blank_enabled = on;
  return DV_OK;
}

// The thresholds the last forward's scan found for its first `n` examples: out[k * n + i], k = 0 rows used (first
// all-zero row), 1 conv2 rows, 2 stem_b rows, 3 the 3x3 80->192's rows, 4 its pooled rows.  Returns
// DV_ERR_UNSUPPORTED when the model does not skip (shape without the uint8 front end, DV_BLANK_SKIP=0, switched off).
int dv_model_blank_thresholds(dv_model* m, int n, int32_t* out) {
  if (!m || !out || n < 1 || n > m->desc.max_batch) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_model_blank_thresholds: bad argument");
  if (!m->blank_on()) return dv::fail(DV_ERR_UNSUPPORTED, "dv_model_blank_thresholds: blank-row skipping is off for this model");
  DV_HIP_CHECK(hipSetDevice(m->device));
  DV_HIP_CHECK(hipDeviceSynchronize());
  for (int k = 0; k < 5; ++k) {
    DV_HIP_CHECK(hipMemcpy(out + static_cast<size_t>(k) * n,
                           static_cast<const int*>(m->d_blank_thr.ptr) + static_cast<size_t>(k) * m->desc.max_batch,
                           static_cast<size_t>(n) * sizeof(int), hipMemcpyDeviceToHost));
  }
  return DV_OK;
}

// Testing hook: how many forwards were captured into a new hipGraph / replayed from the cache.
int dv_model_graph_stats(const dv_model* m, int64_t* captures, int64_t* replays) {
  if (!m) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_model_graph_stats: null");
  if (captures) *captures = m->graph_captures;
  if (replays) *replays = m->graph_replays;
  return DV_OK;
}

}  // extern "C"
