// Device-side pieces shared by the convolution kernels of libdvhip.so (model.hip,
// imgconv.hip): the C8 activation geometry, the launch arguments and the epilogue
// that turns MFMA accumulators into 16-byte pieces of the output tensor(s).
#ifndef DV_CONV_COMMON_H_
#define DV_CONV_COMMON_H_

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace dv {
namespace convk {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));

constexpr int kConvThreads = 256;
constexpr int kChunk = 16;       // channels per K chunk

// Geometry of one activation tensor in HBM: fp16, channel-blocked and
// zero-haloed, [N][C/8][H + 2*halo][W + 2*halo][8].  The halo is written once
// (hipMemset at model creation) and never touched again: producers store the
// interior only, so 'same'-padded convolutions read their padding as ordinary
// in-bounds zeros and need no predicates.
struct TensorGeom {
  int h, w;     // interior size
  int halo;
  int hp, wp;   // padded size
  int groups;   // channel groups of 8 (full concat width)
};

// One output branch of a (possibly grouped) convolution launch.
struct ConvBranch {
  const float* shift;     // [Cout (+pad)] folded BN shift, or NULL (raw output)
  _Float16* out;
  float* out32;           // non-NULL: the output tensor is float32 (same geometry, a piece = 8 floats) -- raw pooled
                          // projections awaiting their average pool, the last block's outputs awaiting the global pool:
                          // tensors no MFMA reads are not rounded to fp16 on the way (round 6)
  TensorGeom og;
  int lo_groups;          // > 0: the output tensor is WIDE (precise mode, model.hip): channel groups [0, lo_groups) hold
                          // hi = fp16(x), groups [lo_groups, 2 lo_groups) hold lo = fp16(x - hi) -- the consumer's K runs over
                          // both with the same weights, i.e. it multiplies 22-bit activations at twice the MFMA count
  int out_goff;           // first destination group of this branch
  int Cout;
  int relu;
  int sub0;               // first 32-cout subtile of this branch in the launch's cout space
  // AveragePooling2D(3, 1, 'same') of this branch's RAW outputs in the launch's epilogue (ConvArgs::tile_g):
  // out / og / out_goff / shift / relu then describe the POOLED tensor (model.hip choose_avg_epilogue)
  int avgpool;
};

constexpr int kMaxBranches = 4;

struct ConvArgs {
  const _Float16* in;
  // Sibling convolutions that read the SAME input with the same geometry (the
  // 1x1 heads of an Inception block) run as one launch over the CONCATENATION of their
  // output channels (each branch padded to whole 32-cout subtiles): one packed weight
  // image, cout tiles of NB*32 that may straddle two branches, and an epilogue that
  // routes every 32-cout subtile to its branch's tensor.  The input is fetched from HBM
  // once and re-read from L2 by ceil(sum couts / (NB*32)) tiles instead of once per
  // branch tile.
  const _Float16* w;      // packed [cout_tile][slab][8 chunks][2 k-groups][NB*32][8]
  ConvBranch br[kMaxBranches];
  int n_branches;
  TensorGeom ig;
  int N, Cin;
  int OH, OW;
  int KH, KW, stride, pad_h, pad_w;
  int M;                  // N*OH*OW
  int n_chunks;
  int n_slabs;            // ceil(n_chunks / kSlabChunks)
  size_t in_bytes;        // size of the input tensor
  unsigned img_bytes;     // bytes of one example of the input tensor (all groups, with halo)
  unsigned chunk_stride;  // bytes between consecutive 16-channel chunks = 2*hp*wp*16
  int n_tiles;            // cout tiles (grid = m_blocks * n_tiles)
  // Row-band mode (band = H > 0): 'same'-padded, stride-1 filters TALLER than the map
  // (7x1 on 4 rows, 3x3 / 3x1 on 1 row).  A block's pixels all lie in ONE output row r, so
  // the taps that fall into the zero halo above / below the map are the same for the whole
  // block and are skipped: KH here is the number of map rows (taps kh = pad_h - r + 0..H-1
  // hit input rows 0..H-1), `w` holds one packed image per output row, and the grid is
  // band * ceil(N*OW / block pixels) * n_tiles.  Skipped products are exact zeros, so the
  // result is bit-identical to the full filter.
  int band;
  // tuning knob (DV_CU_PAIR): renumber blocks so that the two blocks a CU runs concurrently are
  // neighbouring cout tiles of one pixel tile (they then share its L1 lines)
  int cu_pair;
  float rcp_ow, rcp_ohow; // 1/OW, 1/(OH*OW) for the prologue's index split
  // Blank-row skipping (opt-in, DV_BLANK_SKIP; HISTORY.md 7): blank_row[n] = first output row of
  // example n whose receptive field lies entirely in the zero rows below the pile-up.  Such
  // outputs equal the response of the all-blank image at the same position (blank_src: ONE
  // example in the output tensor's geometry), so a block whose pixels all lie there copies
  // instead of multiplying -- bit-identical.  Single-branch launches only; NULL = off.
  const int* blank_row;
  const _Float16* blank_src;
  // blank_need[n] (optional): output rows of example n that the consumer's computed tiles read; a blank block at or
  // below it is not even copied (model.hip blank_need_kernel)
  const int* blank_need;
  // Split weights (model.hip, HISTORY.md 15): the packed image holds every K chunk twice, W_hi then
  // W_lo = fp16(W - W_hi); n_chunks counts both, the pixel operand advances once per pair.
  // split_tiles: the leading cout tiles of the launch that carry such pairs (= n_tiles when every
  // branch is split); the tiles behind them hold plain weights in the first half of their slot
  // (sibling 1x1 heads of which only some are split, HISTORY.md 15).
  int split;
  int split_tiles;
  // Side max-pool (model.hip choose_side_pool, HISTORY.md 4.11): a 3x3 / stride-2 'valid' convolution
  // loads, per 16-channel chunk, exactly the nine pieces of the 3x3 / stride-2 max-pool window of
  // each of its output pixels.  The workgroups of cout tile 0 keep their running maximum and store
  // it -- the sibling MaxPooling2D(3, 2) of the reduction block without its own launch.
  _Float16* side_pool_out;   // NULL = off
  TensorGeom side_pool_og;
  int side_pool_goff;        // first destination group (channel offset / 8) in the concat buffer
  // Image-aligned pixel tiles (tile_g > 0; conv_mfma_kernel<..., AVG>): a 256-pixel block holds tile_g WHOLE
  // maps of tile_p = OH*OW pixels (slots past tile_g * tile_p idle), so that a branch's 3x3 average pool
  // (ConvBranch::avgpool: the pooled projection of an Inception block, evaluated as conv -> pool) never
  // leaves the block: raw outputs go to LDS as fp16, are averaged there, shifted, clamped and stored --
  // the raw tensor and the avg-pool launch disappear.  grid = ceil(N / tile_g) * n_tiles.
  int tile_g, tile_p;
  float rcp_tile_p;
  // Wide input (precise mode, model.hip BufferDesc::wide): the input tensor holds Cin / 8 groups of hi = fp16(x) and,
  // lo_off bytes further in every example, Cin / 8 groups of lo = fp16(x - hi).  Every K chunk is multiplied twice --
  // the same weight fragments against the hi and the lo pixel fragment (conv_slab_wide) -- so the layer sees 22-bit
  // activations for twice the MFMAs at unchanged weight traffic.  n_chunks counts the layer's own chunks.
  int wide_in;
  unsigned lo_off;
};

typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

// q = m / d, r = m % d for 0 <= m < 2^26, d >= 5 via one fp32 multiply + fix-up: float(m)
// is off by <= 2 and the product by a few ulp, so q is off by at most one.
__device__ __forceinline__ void divmod_small(int m, int d, float rcp, int& q, int& r) {
  q = static_cast<int>(static_cast<float>(m) * rcp);
  r = m - q * d;
  if (r < 0) {
    r += d;
    --q;
  }
  if (r >= d) {
    r -= d;
    ++q;
  }
}

// The arithmetic of one average-pool output channel, shared by avgpool3s1_kernel (model.hip) and the
// pooling epilogue below so that both round alike: three column sums (each top + middle + bottom, zeros
// outside the map) of the float32 raw projection, left to right, times 1 / (cells inside), then shift + ReLU
// when the pool carries them.
__device__ __forceinline__ float avg_finish(float c0, float c1, float c2, float inv, float sh, bool shift_relu) {
#pragma clang fp contract(off)   // never (sum * inv + sh) as one fma in one kernel and two roundings in the other
  const float v = (c0 + c1 + c2) * inv;
  return shift_relu ? fmaxf(v + sh, 0.f) : v;
}

// The shifts of one 32-cout subtile as this lane needs them, through the SCALAR cache (constant address
// space, wave-uniform address -> s_load_dwordx8, lgkmcnt) instead of the vector memory queue:
// shv[q][pair] = shifts of couts 8q + 4*hi + {0,1},{2,3}; zeros without a shift array (raw outputs).
__device__ __forceinline__ void load_shifts(const float* shift, int cbase, int hi, float2_t (&shv)[4][2]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), up = lo;
    if (shift != nullptr) {  // uniform; the shift array is padded past Cout
      typedef float f4_t __attribute__((ext_vector_type(4)));
      typedef const f4_t __attribute__((address_space(4))) * const_f4_ptr;
      const_f4_ptr sp = (const_f4_ptr)(reinterpret_cast<uintptr_t>(shift + (cbase + 8 * q)));
      const f4_t l4 = sp[0], u4 = sp[1];
      lo = make_float4(l4[0], l4[1], l4[2], l4[3]);
      up = make_float4(u4[0], u4[1], u4[2], u4[3]);
    }
    shv[q][0] = hi ? float2_t{up.x, up.y} : float2_t{lo.x, lo.y};
    shv[q][1] = hi ? float2_t{up.z, up.w} : float2_t{lo.z, lo.w};
  }
}

// One 32-cout subtile of a wave tile -> its branch's tensor: shift + ReLU, then either fp16 -- lanes l / l+32
// pair their halves into 16-byte pieces (v_permlane32_swap), 32 consecutive pixels = one 512-byte run -- or
// float32 (ConvBranch::out32: each lane stores its four consecutive couts of every group, 16 bytes).
template <int PT>
__device__ __forceinline__ void epilogue_subtile(const float16_t (&acc)[PT], const ConvBranch& b, int cbase,
                                                 const int (&pn)[PT], const int (&poh)[PT], const int (&pow_)[PT],
                                                 const bool (&mvalid)[PT], int lane) {
  const int hi = lane >> 5;
  const half2_t zero2 = {static_cast<_Float16>(0.f), static_cast<_Float16>(0.f)};
  const unsigned gstride = static_cast<unsigned>(b.og.hp * b.og.wp);
  float2_t shv[4][2];
  load_shifts(b.shift, cbase, hi, shv);
  if (b.out32 != nullptr) {   // wave-uniform
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const float16_t a = acc[pt];
      const unsigned obase = static_cast<unsigned>(
          ((pn[pt] * b.og.groups + b.out_goff) * b.og.hp + poh[pt] + b.og.halo) * b.og.wp + pow_[pt] + b.og.halo);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 v = make_float4(a[4 * q] + shv[q][0][0], a[4 * q + 1] + shv[q][0][1], a[4 * q + 2] + shv[q][1][0],
                               a[4 * q + 3] + shv[q][1][1]);
        if (b.relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        const int group = cbase / 8 + q;
        if (mvalid[pt] && group * 8 < b.Cout) {
          *reinterpret_cast<float4*>(b.out32 + (static_cast<size_t>(obase) + static_cast<size_t>(group) * gstride) * 8 +
                                     4 * hi) = v;
        }
      }
    }
    return;
  }
  uint4_t* outp = reinterpret_cast<uint4_t*>(b.out);
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const float16_t a = acc[pt];
    // piece index of (n, group out_goff, oh, ow) in this branch's output tensor
    const unsigned obase = static_cast<unsigned>(
        ((pn[pt] * b.og.groups + b.out_goff) * b.og.hp + poh[pt] + b.og.halo) * b.og.wp +
        pow_[pt] + b.og.halo);
    unsigned pk[4][2];  // [q][dword]: 4 halfs of group q held by this lane
    unsigned pl[4][2];  // the same for the lo pieces of a wide output
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int hq = 0; hq < 2; ++hq) {
        const float2_t v = float2_t{a[4 * q + 2 * hq], a[4 * q + 2 * hq + 1]} + shv[q][hq];
        half2_t h = __builtin_convertvector(v, half2_t);
        if (b.relu) h = __builtin_elementwise_max(h, zero2);
        pk[q][hq] = __builtin_bit_cast(unsigned, h);
        if (b.lo_groups > 0) {   // wave-uniform
          const float2_t x = b.relu ? float2_t{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)} : v;
          const float2_t r = x - __builtin_convertvector(h, float2_t);
          pl[q][hq] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, half2_t));
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      // v_permlane32_swap(x, y): x' = {x.lo, y.lo}, y' = {x.hi, y.hi}.  With
      // x = group 2t and y = group 2t+1, {x', y'} is the full 8-cout piece of
      // group 2t in the low half-wave and of group 2t+1 in the high one.
      const auto d0 = __builtin_amdgcn_permlane32_swap(pk[2 * t][0], pk[2 * t + 1][0], false, false);
      const auto d1 = __builtin_amdgcn_permlane32_swap(pk[2 * t][1], pk[2 * t + 1][1], false, false);
      const uint4_t piece = {d0[0], d1[0], d0[1], d1[1]};
      const int group = cbase / 8 + 2 * t + hi;
#ifdef DV_ABLATE_EPI    // timing ablation (tools/ablate_conv.sh): store only a value that never occurs
      if (mvalid[pt] && group * 8 < b.Cout && piece[0] == 0x7e017e01u) {
#else
      if (mvalid[pt] && group * 8 < b.Cout) {
#endif
        outp[obase + static_cast<unsigned>(group) * gstride] = piece;
      }
      if (b.lo_groups > 0) {
        const auto e0 = __builtin_amdgcn_permlane32_swap(pl[2 * t][0], pl[2 * t + 1][0], false, false);
        const auto e1 = __builtin_amdgcn_permlane32_swap(pl[2 * t][1], pl[2 * t + 1][1], false, false);
        const uint4_t lo_piece = {e0[0], e1[0], e0[1], e1[1]};
        if (mvalid[pt] && group * 8 < b.Cout) {
          outp[obase + static_cast<unsigned>(group + b.lo_groups) * gstride] = lo_piece;
        }
      }
    }
  }
}

// branch of 32-cout subtile `sub` of the launch (wave-uniform; the branch table sits in the kernarg
// segment and is indexed with scalar loads)
__device__ __forceinline__ int branch_of(const ConvArgs& p, int sub) {
  int bi = 0;
#pragma unroll
  for (int i = 1; i < kMaxBranches; ++i) bi += (i < p.n_branches && sub >= p.br[i].sub0) ? 1 : 0;
  return bi;
}

// Epilogue of one wave tile.
template <int NB, int PT>
__device__ __forceinline__ void conv_epilogue(const float16_t (&acc)[NB][PT], const ConvArgs& p,
                                              int n_tile, const int (&pn)[PT], const int (&poh)[PT],
                                              const int (&pow_)[PT], const bool (&mvalid)[PT],
                                              int lane) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int sub = n_tile * NB + nb;
    const ConvBranch& b = p.br[branch_of(p, sub)];
    const int cbase = (sub - b.sub0) * 32;  // first cout of the subtile within its branch
    if (cbase >= b.Cout) continue;           // padding subtile past the last branch
    epilogue_subtile<PT>(acc[nb], b, cbase, pn, poh, pow_, mvalid, lane);
  }
}

// Epilogue of conv_mfma_kernel<..., AVG>: branches without `avgpool` store as conv_epilogue does; the others
// leave their RAW float32 outputs in LDS (the weight slabs' space holds two 32-cout subtiles x 256 pixel slots
// x 4 bytes at a time, so a 128-cout tile is pooled in two halves), [subtile][8-cout group q][pixel slot]
// [8 floats], and after a barrier every thread averages the 3x3 neighbourhood of ITS pixel slot (the block holds
// whole maps, ConvArgs::tile_g) for the 4 groups of each pooled subtile: avgpool3s1_kernel's arithmetic
// (avg_finish), shift, ReLU, one store per group into the pooled tensor.  Bit-identical to
// conv (float32 raw tensor) -> avgpool3s1_kernel.  Round 6: the raw projection is no longer rounded to fp16
// before it is averaged (the reference pools in float32; deepvariant/call_variants.py:913-918).
template <int NB, int PT>
__device__ __forceinline__ void conv_epilogue_avg(const float16_t (&acc)[NB][PT], const ConvArgs& p, int n_tile,
                                                  int pix_block, const int (&pn)[PT], const int (&poh)[PT],
                                                  const int (&pow_)[PT], const bool (&mvalid)[PT], int lane,
                                                  int wave, _Float16* smem) {
  static_assert(PT == 2 && NB % 2 == 0, "256-pixel blocks, subtiles pooled in pairs");
  const int hi = lane >> 5;
  float* tile = reinterpret_cast<float*>(smem);
  // the pooling pass's pixel: thread = pixel slot
  const int my_slot = wave * 64 + lane;
  int il, pix, y, x;
  divmod_small(my_slot, p.tile_p, p.rcp_tile_p, il, pix);
  divmod_small(pix, p.OW, p.rcp_ow, y, x);
  const int n = pix_block * p.tile_g + il;
  const bool live = il < p.tile_g && n < p.N;
  const int H = p.OH, W = p.OW;
  const int map0 = il * p.tile_p;
  const float inv = 1.0f / static_cast<float>(((y > 0) + (y < H - 1) + 1) * ((x > 0) + (x < W - 1) + 1));
#pragma unroll
  for (int half = 0; half < NB / 2; ++half) {
    bool any_pooled = false;   // block-uniform
#pragma unroll
    for (int nbl = 0; nbl < 2; ++nbl) {
      const int sub = n_tile * NB + 2 * half + nbl;
      const ConvBranch& b = p.br[branch_of(p, sub)];
      const int cbase = (sub - b.sub0) * 32;
      if (cbase >= b.Cout) continue;
      if (b.avgpool == 0) {
        epilogue_subtile<PT>(acc[2 * half + nbl], b, cbase, pn, poh, pow_, mvalid, lane);
      } else {
        any_pooled = true;
      }
    }
    if (!any_pooled) continue;
    __syncthreads();   // every wave has read its last weight fragment / finished the previous half's pooling pass
#pragma unroll
    for (int nbl = 0; nbl < 2; ++nbl) {
      const int sub = n_tile * NB + 2 * half + nbl;
      const ConvBranch& b = p.br[branch_of(p, sub)];
      const int cbase = (sub - b.sub0) * 32;
      if (cbase >= b.Cout || b.avgpool == 0) continue;
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const float16_t a = acc[2 * half + nbl][pt];
        const int slot = (wave * PT + pt) * 32 + (lane & 31);
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // idle slots too: never read.  [subtile][group q][half hi][slot][4 floats]: a half-
          // wave's 16-byte stores are consecutive (the [slot][8 floats] image cost 4-way bank conflicts)
          reinterpret_cast<float4*>(tile)[((nbl * 4 + q) * 2 + hi) * 256 + slot] =
              make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int nbl = 0; nbl < 2; ++nbl) {
      const int sub = n_tile * NB + 2 * half + nbl;
      const ConvBranch& b = p.br[branch_of(p, sub)];
      const int cbase = (sub - b.sub0) * 32;
      if (cbase >= b.Cout || b.avgpool == 0 || !live) continue;
      const unsigned gstride = static_cast<unsigned>(b.og.hp * b.og.wp);
      const unsigned obase = static_cast<unsigned>(
          ((n * b.og.groups + b.out_goff) * b.og.hp + y + b.og.halo) * b.og.wp + x + b.og.halo);
      for (int q8 = 0; q8 < 4; ++q8) {
        const int group = cbase / 8 + q8;
        if (group * 8 >= b.Cout) break;
        const float4* src = reinterpret_cast<const float4*>(tile) + (nbl * 4 + q8) * 2 * 256 + map0;
        // Separable: every thread sums ITS column (rows y-1 .. y+1 of column x: three reads) and takes the two
        // neighbouring columns' sums from lanes -1 / +1 (v_mov_dpp wave_shr / wave_shl: no LDS) -- consecutive slots of a
        // map are consecutive columns of a row.  The first and last lane of a wave fetch the column their missing
        // neighbour would have summed themselves (two active lanes: no LDS bandwidth to speak of).  Round 6: a third of
        // the LDS reads of the nine-point form, whose 72 ds_read_b128 per thread and pooled subtile made the pooling
        // epilogue LDS-bound (profiles/r06_experiments.txt).  Same sums in the same order as avgpool3s1_kernel.
        auto column = [&](int xc, bool wanted, float (&v)[8]) {
          float r[3][8];
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int yy = y + dy - 1;
            const bool ok = wanted && yy >= 0 && yy < H;
            float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), up = lo;
            if (ok) {
              lo = src[yy * W + xc];
              up = src[256 + yy * W + xc];
            }
            r[dy][0] = lo.x; r[dy][1] = lo.y; r[dy][2] = lo.z; r[dy][3] = lo.w;
            r[dy][4] = up.x; r[dy][5] = up.y; r[dy][6] = up.z; r[dy][7] = up.w;
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = r[0][j] + r[1][j] + r[2][j];
        };
        float col[3][8];
        column(x, true, col[1]);
        const bool first = lane == 0, last = lane == 63;
        const int xe = first ? x - 1 : x + 1;
        float edge[8];
        column(xe, (first || last) && xe >= 0 && xe < W, edge);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int vi = __builtin_bit_cast(int, col[1][j]);
          const float from_left = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, vi, 0x138, 0xf, 0xf, false));   // wave_shr:1
          const float from_right = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, vi, 0x130, 0xf, 0xf, false));  // wave_shl:1
          col[0][j] = x == 0 ? 0.f : first ? edge[j] : from_left;
          col[2][j] = x == W - 1 ? 0.f : last ? edge[j] : from_right;
        }
        float sh[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (b.shift != nullptr) {
          const float4 s0 = *reinterpret_cast<const float4*>(b.shift + group * 8);
          const float4 s1 = *reinterpret_cast<const float4*>(b.shift + group * 8 + 4);
          sh[0] = s0.x; sh[1] = s0.y; sh[2] = s0.z; sh[3] = s0.w;
          sh[4] = s1.x; sh[5] = s1.y; sh[6] = s1.z; sh[7] = s1.w;
        }
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = avg_finish(col[0][j], col[1][j], col[2][j], inv, sh[j], b.shift != nullptr);
        const size_t at = static_cast<size_t>(obase) + static_cast<size_t>(group) * gstride;
        if (b.out32 != nullptr) {
          float4* d = reinterpret_cast<float4*>(b.out32 + at * 8);
          d[0] = make_float4(o[0], o[1], o[2], o[3]);
          d[1] = make_float4(o[4], o[5], o[6], o[7]);
        } else {
          half8_t h;
#pragma unroll
          for (int j = 0; j < 8; ++j) h[j] = static_cast<_Float16>(o[j]);
          reinterpret_cast<uint4_t*>(b.out)[at] = __builtin_bit_cast(uint4_t, h);
          if (b.lo_groups > 0) {
            half8_t l;
#pragma unroll
            for (int j = 0; j < 8; ++j) l[j] = static_cast<_Float16>(o[j] - static_cast<float>(h[j]));
            reinterpret_cast<uint4_t*>(b.out)[at + static_cast<size_t>(b.lo_groups) * gstride] = __builtin_bit_cast(uint4_t, l);
          }
        }
      }
    }
  }
}

}  // namespace convk
}  // namespace dv

#endif  // DV_CONV_COMMON_H_
