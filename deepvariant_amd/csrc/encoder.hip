// encoder.hip -- pileup-image encoder for gfx950 (MI355X).
//
// One workgroup (4 wave64) builds one item = one
// PileupImageEncoderNative::BuildPileupForOneSample call of the reference
// (deepvariant/pileup_image_native.cc:297-447) and streams it out in the HWC
// byte layout of FillPileupArray (deepvariant/pileup_image_native.h:214-275):
//
//   phase A  one thread per (shuffled) list entry: mapq gate
//            (pileup_image_native.cc:485-491) and the low-base-quality-at-the-
//            variant-start gate (pileup_channel_lib.cc:142-148) by walking the
//            read's CIGAR (pileup_channel_lib.cc:213-260);
//   phase B  block prefix sum over the accept flags: the first
//            height - reference_band_height accepted entries are kept
//            (pileup_image_native.cc:367-403);
//   phase C  stable rank sort of <= 256 kept rows by (hap, allele group,
//            position, name rank) (pileup_image_native.cc:75-102,405);
//   phase D  each wave renders whole rows: every lane resolves what the read
//            puts in its column (last CIGAR event wins, as in the reference's
//            in-order overwrites), channel bytes come from host-built LUTs
//            (bit-exact by construction: the fp32 truncations of
//            channels/*.cc are evaluated once on the host), the row is staged
//            HWC in LDS at the byte phase of its global address and leaves as
//            fully coalesced aligned dword stores; rows below the pileup are
//            zero-filled with 16-byte stores.
//
// The kernel is HBM-write bound: ~155 KB written per item
// against ~12 KB of packed reads, most of which hit L2 because neighbouring
// candidates share reads.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>

#include <type_traits>

#include "dv_internal.h"

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kColsPerLane = 4;  // a row is rendered in passes of 256 columns, 4 per lane
constexpr int kMaxKept = 256;
constexpr int kCigCache = 8;     // CIGAR words per kept read cached in LDS when the batch gives no hint (EncArgs::cig_cache:
                                 // a power of two, 8..64, from dv_batch::max_cigar_ops; longer CIGARs are read from global)
constexpr int kCigCacheMax = 16; // by default; up to kCigCacheHardMax words (one coalesced wave load per read) fit the code
constexpr int kCigCacheHardMax = 64;   // (DV_CIG_CACHE: measured on ont50, 13,654 images: 8 words 1.94 ms, 16: 1.92, 32: 2.00, 64: 2.46 --
                                       // the LDS the larger tables take costs more occupancy than the cached words save)
constexpr int kPixDw = DV_MAX_CHANNELS / 4;  // dwords of one pixel's channel bytes
constexpr int kInsertLutSize = 1008;

enum ChannelKind : uint8_t {
  kZero = 0,      // blank / mean_coverage (painted afterwards)
  kBase,          // LUT on the read base
  kBaseQual,      // LUT on the base quality
  kMapq,          // LUT on the mapping quality (per read)
  kStrand,        // per read
  kSupport,       // LUT on list_code (per item,read)
  kDiff,          // read base == ref base
  kInsert,        // LUT on |fragment_length| (per read)
  kHaplotype,     // per read
  kSupplementary, // per read
  kMeth5,         // LUT on the 5mC byte of the base
  kMeth6,         // LUT on the 6mA byte of the base
  kAuxRead0,      // host-computed per-read pixel, read_aux[k], k = kind - kAuxRead0
  kAuxRead1,
  kAuxRead2,
  kAuxRead3,
  kAuxRead4,      // gc_content of the read (read_aux[4]); reference row: ref_aux2
  kAuxList,       // host-computed per-(item,read) pixel, list_aux
  kBaseAux0,      // host-computed per-base pixel (is_homopolymer); reference row: ref_aux0
  kBaseAux1,      // host-computed per-base pixel (homopolymer_weighted); reference row: ref_aux1
  kBaseAux2,      // host-computed per-base pixel, third plane (flow-space channels; these may also sit in planes 0 / 1)
};

// Everything the kernel needs besides the batch; copied into LDS per workgroup.
struct alignas(16) EncConst {
  uint8_t lut_base[256];
  uint8_t lut_bq[256];
  uint8_t lut_mapq[256];
  uint8_t lut_mod[256];
  uint8_t lut_insert[kInsertLutSize];
  uint8_t lut_support[4];
  uint8_t strand[2];  // [reverse]
  uint8_t diff[2];    // [matches]
  uint8_t supp[2];    // [is supplementary]
  uint8_t pad0[6];
  uint8_t kind[DV_MAX_CHANNELS];
  uint8_t ref_const[DV_MAX_CHANNELS];
  uint32_t ref_konst[DV_MAX_CHANNELS / 4];  // reference-row pixel: constants, 4 channels per dword
  uint32_t ref_sel[DV_MAX_CHANNELS / 4];    // v_perm selectors: 4 = base colour, i = constant i
  int32_t n_channels;
  int32_t width;
  int32_t band;
  int32_t min_bq;
  int32_t min_mapq;
  int32_t anchor_char;
  int32_t sort_by_haplotypes;
  int32_t polishing;
  int32_t sort_by_group;
  int32_t mean_cov_channel;  // index or -1
  int32_t pad1[2];
};
static_assert(sizeof(EncConst) % 16 == 0, "EncConst must be 16-byte sized");

struct EncArgs {
  const EncConst* konst;
  const uint32_t* perm_off;
  const uint16_t* perm;
  // reads
  const int32_t* read_pos;
  const int32_t* read_sort_pos;
  const uint32_t* read_seq_off;
  const uint32_t* read_cigar_off;
  const uint8_t* read_mapq;
  const uint8_t* read_flags;
  const int32_t* read_frag_len;
  const int32_t* read_hp;
  const uint32_t* read_name_rank;
  const uint8_t* read_aux;
  const uint8_t* bases;
  const uint8_t* quals;
  const uint8_t* mod_5mc;
  const uint8_t* mod_6ma;
  const uint32_t* cigar;
  // items
  const int32_t* item_variant_start;
  const int32_t* item_image_start;
  const uint32_t* item_ref_idx;
  const uint32_t* item_list_off;
  const uint16_t* item_height;
  const uint64_t* item_out_off;
  const uint32_t* item_blank_mask;
  const float* item_mean_coverage;
  const uint8_t* ref_windows;
  const uint32_t* list_read;
  const uint8_t* list_code;
  const uint8_t* list_group;
  const uint8_t* list_aux;
  const uint8_t* base_aux0;
  const uint8_t* base_aux1;
  const uint8_t* base_aux2;
  const uint8_t* ref_aux0;
  const uint8_t* ref_aux1;
  const uint8_t* ref_aux2;
  int32_t n_items;
  int32_t n_channels;     // = EncConst::n_channels (sizes the LDS layout)
  int32_t out_channels;
  int32_t row_buf_bytes;  // per-wave LDS row buffer, multiple of 16
  int32_t cig_cache;      // CIGAR words per kept read in LDS: power of two in [kCigCache, kCigCacheMax]
  int32_t kept_cap;       // rows of the CIGAR cache: max (item_height - band) of the batch, <= kMaxKept
  uint8_t* out;
  int32_t* out_rows;
};

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// CalculateBaseLevelData restricted to the events that land on ref position
// `vstart` (pileup_channel_lib.cc:126-165,213-260): true = keep the read.
__device__ bool read_passes_site_gate(const EncArgs& a, const EncConst& c,
                                      uint32_t r, int vstart, int istart) {
  const int vcol = vstart - istart;
  if (vcol < 0 || vcol >= c.width) return true;  // `col < ref_bases.size()` never holds
  const uint32_t c0 = a.read_cigar_off[r], c1 = a.read_cigar_off[r + 1];
  const uint32_t s0 = a.read_seq_off[r];
  int ref_i = a.read_pos[r];
  int read_i = 0;
  for (uint32_t k = c0; k < c1; ++k) {
    const uint32_t cg = a.cigar[k];
    const int op = cg & 0xF;
    const int len = cg >> 4;
    switch (op) {
      case DV_CIGAR_ALIGNMENT_MATCH:
      case DV_CIGAR_SEQUENCE_MATCH:
      case DV_CIGAR_SEQUENCE_MISMATCH:
        if (vstart >= ref_i && vstart < ref_i + len) {
          // a NUL base would not be drawn (`read_base &&`), never in practice
          const uint32_t idx = s0 + read_i + (vstart - ref_i);
          if (a.bases[idx] != 0 && a.quals[idx] < c.min_bq) return false;
        }
        ref_i += len;
        read_i += len;
        break;
      case DV_CIGAR_INSERT:
        if (ref_i > 0 && ref_i - 1 == vstart && c.anchor_char != 0) {
          if (a.quals[s0 + read_i] < c.min_bq) return false;
        }
        read_i += len;
        break;
      case DV_CIGAR_CLIP_SOFT:
        read_i += len;
        break;
      case DV_CIGAR_DELETE:
        if (read_i > 0 && ref_i - 1 == vstart && c.anchor_char != 0) {
          if (a.quals[s0 + read_i - 1] < c.min_bq) return false;
        }
        ref_i += len;
        break;
      case DV_CIGAR_SKIP:
        ref_i += len;
        break;
      default:
        break;
    }
  }
  return true;
}

// Writes one pixel's CO channel bytes (packed 4 per dword in o[]) at an arbitrary byte
// address of the LDS row buffer.
__device__ __forceinline__ void store_pixel(uint8_t* dst, const uint32_t (&o)[kPixDw], int CO) {
  if (CO == 7) {
    __builtin_memcpy(dst, &o[0], 4);
    const uint16_t m = static_cast<uint16_t>(o[1]);
    __builtin_memcpy(dst + 4, &m, 2);
    dst[6] = static_cast<uint8_t>(o[1] >> 16);
  } else if (CO == 6) {
    __builtin_memcpy(dst, &o[0], 4);
    const uint16_t m = static_cast<uint16_t>(o[1]);
    __builtin_memcpy(dst + 4, &m, 2);
  } else {
    for (int ch = 0; ch < CO; ++ch) dst[ch] = static_cast<uint8_t>(o[ch >> 2] >> ((ch & 3) * 8));
  }
}

// HaplotypeTagChannel (channels/haplotype_tag_channel.cc:76-110).
__device__ __forceinline__ uint8_t haplotype_pixel(int hp, int polishing) {
  int v = (hp == DV_HP_NONE) ? 0 : hp;
  if (polishing == 2) {
    if (v == 1) v = 2;
    else if (v == 2) v = 1;
  }
  if (static_cast<float>(v) > 2.0f) v = 2;
  return static_cast<uint8_t>(
      static_cast<int>(254.0f * (static_cast<float>(v) / 2.0f)));
}

// GetHapIndex (pileup_image_native.cc:449-475).
__device__ __forceinline__ int hap_index(int hp, const EncConst& c) {
  if (!c.sort_by_haplotypes || hp == DV_HP_NONE) return 0;
  if (c.polishing > 0 && hp == c.polishing) return -1;
  if (hp < 0) return 0;
  return hp;
}

__global__ __launch_bounds__(kBlock) void encode_items_kernel(EncArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  EncConst* c = reinterpret_cast<EncConst*>(smem);
  uint32_t* kept_read = reinterpret_cast<uint32_t*>(smem + sizeof(EncConst));
  uint32_t* kept_src = kept_read + kMaxKept;
  int32_t* key_hap = reinterpret_cast<int32_t*>(kept_src + kMaxKept);
  int32_t* key_pos = key_hap + kMaxKept;
  uint32_t* key_rank = reinterpret_cast<uint32_t*>(key_pos + kMaxKept);
  uint32_t* order = key_rank + kMaxKept;
  uint32_t* wave_tot = order + kMaxKept;          // [kWaves]
  // per kept read: pixel program = constant bytes + v_perm selectors, 4 channels per dword
  const int pdw = (a.n_channels + 3) >> 2;         // dwords per pixel program entry
  uint32_t* m_const = wave_tot + 8;                // [kept_cap][pdw] (kept <= max_reads <= kept_cap)
  uint32_t* m_sel_a = m_const + a.kept_cap * pdw;  // dynamic bytes {base, qual, diff, 5mC}
  uint32_t* m_sel_b = m_sel_a + a.kept_cap * pdw;  // dynamic byte  {6mA}
  const int cig_cache = a.cig_cache;
  uint32_t* m_cig = m_sel_b + a.kept_cap * pdw;    // [kept_cap][cig_cache]: the first CIGAR words of each kept read
  uint8_t* row_bufs = reinterpret_cast<uint8_t*>(m_cig + a.kept_cap * cig_cache);
  // after the sort the key arrays are dead: they become the per-read metadata
  uint32_t* m_c0 = reinterpret_cast<uint32_t*>(key_hap);
  uint32_t* m_s0 = reinterpret_cast<uint32_t*>(key_pos);
  int32_t* m_rpos = reinterpret_cast<int32_t*>(key_rank);
  uint32_t* m_c1 = kept_src;
  uint32_t* m_flags = kept_read;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int item = blockIdx.x;

  if (tid == 0) wave_tot[kWaves] = 0;   // "some kept read's CIGAR does not fit the LDS cache" (set in phase C')
  {  // constants -> LDS (16-byte copies)
    const uint4* src = reinterpret_cast<const uint4*>(a.konst);
    uint4* dst = reinterpret_cast<uint4*>(c);
    for (int i = tid; i < static_cast<int>(sizeof(EncConst) / 16); i += kBlock) {
      dst[i] = src[i];
    }
  }
  __syncthreads();

  const int W = c->width;
  const int C = c->n_channels;
  const int CO = a.out_channels;
  const int band = c->band;
  const int H = a.item_height[item];
  // host batches are validated (height - band <= kMaxKept); device batches are clamped so
  // that a bad height can never run past the LDS tables
  const int max_reads = min(H - band, a.kept_cap);
  const uint32_t l0 = a.item_list_off[item];
  const int n = static_cast<int>(a.item_list_off[item + 1] - l0);
  const int vstart = a.item_variant_start[item];
  const int istart = a.item_image_start[item];
  const uint32_t blank_mask = a.item_blank_mask ? a.item_blank_mask[item] : 0u;
  const bool shuffled = n > max_reads;  // DownsampleReadIndices
  const uint16_t* perm = shuffled ? a.perm + a.perm_off[n] : nullptr;

  // ---------------- phases A + B: accept flags, keep the first max_reads ----
  int kept = 0;
  for (int cb = 0; cb < n && kept < max_reads; cb += kBlock) {
    const int e = cb + tid;
    bool accept = false;
    uint32_t src = 0, r = 0;
    if (e < n) {
      src = shuffled ? perm[e] : static_cast<uint32_t>(e);
      r = a.list_read[l0 + src];
      accept = a.read_mapq[r] >= c->min_mapq &&
               read_passes_site_gate(a, *c, r, vstart, istart);
    }
    const unsigned long long ballot = __ballot(accept);
    const int before = __popcll(ballot & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(ballot);
    __syncthreads();
    int prefix = before;
    int total = 0;
    for (int w = 0; w < kWaves; ++w) {
      if (w < wave) prefix += wave_tot[w];
      total += wave_tot[w];
    }
    const int slot = kept + prefix;
    if (accept && slot < max_reads) {
      kept_read[slot] = r;
      kept_src[slot] = src;
    }
    kept = min(kept + total, max_reads);
    __syncthreads();
  }

  // ---------------- phase C: stable rank sort of the kept rows --------------
  if (tid < kept) {
    const uint32_t r = kept_read[tid];
    const int hap = hap_index(a.read_hp[r], *c);
    const int group = (c->sort_by_group && a.list_group)
                          ? a.list_group[l0 + kept_src[tid]]
                          : 0;
    // (hap, group) compare lexicographically; hap >= -1 and group < 256.
    key_hap[tid] = hap * 256 + group;
    key_pos[tid] = a.read_sort_pos ? a.read_sort_pos[r] : a.read_pos[r];
    key_rank[tid] = a.read_name_rank[r];
  }
  __syncthreads();
  if (tid < kept) {
    const int h = key_hap[tid], p = key_pos[tid];
    const uint32_t nr = key_rank[tid];
    int rank = 0;
    for (int j = 0; j < kept; ++j) {
      const int hj = key_hap[j], pj = key_pos[j];
      const uint32_t nj = key_rank[j];
      const bool less = (hj != h) ? (hj < h)
                        : (pj != p) ? (pj < p)
                        : (nj != nr) ? (nj < nr)
                                     : (j < tid);
      rank += less ? 1 : 0;
    }
    order[rank] = tid;
  }
  __syncthreads();

  // ---------------- phase C': per-read metadata + pixel program ---------------
  // One thread per kept read fetches everything phase D needs about it (one memory
  // round trip for all rows instead of one per row) and compiles its pixel: the bytes
  // that are constant along the read (mapq, strand, support, insert size, ...) and, per
  // channel, a v_perm_b32 selector that either keeps that constant or picks one of the
  // per-base bytes {base, quality, differs-from-ref, 5mC | 6mA}.
  if (tid < kept) {
    const uint32_t r = kept_read[tid];
    const uint32_t le = l0 + kept_src[tid];
    const uint8_t flags = a.read_flags[r];
    uint32_t konst[kPixDw], sel_a[kPixDw], sel_b[kPixDw];
#pragma unroll
    for (int d = 0; d < kPixDw; ++d) {
      konst[d] = 0;
      sel_a[d] = 0x03020100u;  // identity: keep the bytes of the second operand
      sel_b[d] = 0x03020100u;
    }
    for (int ch = 0; ch < C; ++ch) {
      if ((blank_mask >> ch) & 1u) continue;
      const int d = ch >> 2, sh = (ch & 3) * 8;
      uint32_t v = 0;
      int dyn_a = -1, dyn_b = -1;
      switch (c->kind[ch]) {
        case kBase: dyn_a = 0; break;
        case kBaseQual: dyn_a = 1; break;
        case kDiff: dyn_a = 2; break;
        case kMeth5: if ((flags & DV_READ_HAS_5MC) && a.mod_5mc) dyn_a = 3; break;
        case kMeth6: if ((flags & DV_READ_HAS_6MA) && a.mod_6ma) dyn_b = 0; break;
        case kMapq: v = c->lut_mapq[a.read_mapq[r]]; break;
        case kStrand: v = c->strand[flags & DV_READ_REVERSE ? 1 : 0]; break;
        case kSupport: v = c->lut_support[min<int>(a.list_code[le], 3)]; break;
        case kInsert: {
          int f = a.read_frag_len[r];
          f = f < 0 ? -f : f;
          v = c->lut_insert[min(f, 1000)];
          break;
        }
        case kHaplotype: v = haplotype_pixel(a.read_hp[r], c->polishing); break;
        case kSupplementary: v = c->supp[flags & DV_READ_SUPPLEMENTARY ? 1 : 0]; break;
        case kAuxRead0: case kAuxRead1: case kAuxRead2: case kAuxRead3: case kAuxRead4:
          v = a.read_aux ? a.read_aux[static_cast<size_t>(r) * DV_READ_AUX_STRIDE +
                                      (c->kind[ch] - kAuxRead0)]
                         : 0;
          break;
        case kAuxList: v = a.list_aux ? a.list_aux[le] : 0; break;
        case kBaseAux0: if (a.base_aux0) dyn_b = 1; break;
        case kBaseAux1: if (a.base_aux1) dyn_b = 2; break;
        case kBaseAux2: if (a.base_aux2) dyn_b = 3; break;
        default: break;  // kZero
      }
      konst[d] |= v << sh;
      if (dyn_a >= 0) sel_a[d] = (sel_a[d] & ~(0xFFu << sh)) | (static_cast<uint32_t>(4 + dyn_a) << sh);
      if (dyn_b >= 0) sel_b[d] = (sel_b[d] & ~(0xFFu << sh)) | (static_cast<uint32_t>(4 + dyn_b) << sh);
    }
    const uint32_t rc0 = a.read_cigar_off[r], rc1 = a.read_cigar_off[r + 1];
    const uint32_t rs0 = a.read_seq_off[r];
    const int rp = a.read_pos[r];
#pragma unroll
    for (int d = 0; d < kPixDw; ++d) {
      if (d < pdw) {
        m_const[tid * pdw + d] = konst[d];
        m_sel_a[tid * pdw + d] = sel_a[d];
        m_sel_b[tid * pdw + d] = sel_b[d];
      }
    }
    if (rc1 - rc0 > static_cast<uint32_t>(cig_cache)) atomicOr(&wave_tot[kWaves], 1u);
    m_c0[tid] = rc0;   // (key_* / kept_* of this slot are dead from here on)
    m_s0[tid] = rs0;
    m_rpos[tid] = rp;
    m_c1[tid] = rc1;
    m_flags[tid] = flags;
  }
  __syncthreads();
  // the CIGAR cache: a wave per kept read, its first cig_cache words with ONE coalesced load
  // (cig_cache <= 64 lanes), zero past the read's own operations
  for (int slot = wave; slot < kept; slot += kWaves) {
    const uint32_t rc0 = m_c0[slot], rc1 = m_c1[slot];
    if (lane < cig_cache) m_cig[slot * cig_cache + lane] = rc0 + lane < rc1 ? a.cigar[rc0 + lane] : 0u;
  }
  __syncthreads();

  // ---------------- phase D: render rows -------------------------------------
  const bool any_long_cigar = wave_tot[kWaves] != 0;
  const int row_bytes = W * CO;
  const uint64_t out0 = a.item_out_off[item];
  int mc_limit = 0;  // rows [0, mc_limit) get the mean-coverage paint
  if (c->mean_cov_channel >= 0) {
    const float mc = a.item_mean_coverage ? a.item_mean_coverage[item] : 0.0f;
    mc_limit = min(static_cast<int>(mc) + band, H);
  }
  const int n_render = min(max(band + kept, mc_limit), H);
  uint8_t* rb = row_bufs + wave * a.row_buf_bytes;
  const uint8_t* ref = a.ref_windows + static_cast<size_t>(a.item_ref_idx[item]) * W;
  const int n_pix_dw = pdw;

  // The row is staged in LDS at the byte phase of its global address so that it leaves as
  // aligned 16-byte pieces; `zero` clears the buffer first (rows that do not write every pixel).
  auto row_begin = [&](int row, bool zero) -> uint8_t* {
    const uint64_t gaddr = reinterpret_cast<uint64_t>(a.out) + out0 + static_cast<uint64_t>(row) * row_bytes;
    const int s = static_cast<int>(gaddr & 15);
    if (zero) {
      const int ndw = (s + row_bytes + 3) >> 2;
      uint32_t* rb32 = reinterpret_cast<uint32_t*>(rb);
      for (int k = lane; k < ndw; k += 64) rb32[k] = 0;
    }
    wave_sync();
    return rb + s;
  };
  // LDS row -> HBM: aligned 16-byte pieces; the (<= 15 byte) ragged ends of the row go out as
  // byte stores so neighbouring rows never race on a piece.
  auto row_flush = [&](int row) {
    wave_sync();
    const uint64_t gaddr = reinterpret_cast<uint64_t>(a.out) + out0 + static_cast<uint64_t>(row) * row_bytes;
    const int s = static_cast<int>(gaddr & 15);
    uint8_t* gbase = reinterpret_cast<uint8_t*>(gaddr - s);
    const int npc = (s + row_bytes + 15) >> 4;
    const uint4* rb128 = reinterpret_cast<const uint4*>(rb);
    for (int k = lane; k < npc; k += 64) {
      const int lo = 16 * k - s;
      if (lo >= 0 && lo + 16 <= row_bytes) {
        *reinterpret_cast<uint4*>(gbase + 16 * k) = rb128[k];
      }
    }
    // head: bytes [0, head) of the row; tail: bytes [tail0, row_bytes)
    const int head = min((16 - s) & 15, row_bytes);
    const int tail0 = max(head, ((s + row_bytes) & ~15) - s);
    if (lane < head) gbase[s + lane] = rb[s + lane];
    const int t = tail0 + lane - 16;  // lanes 16..31 take the tail
    if (lane >= 16 && lane < 32 && t < row_bytes) gbase[s + t] = rb[s + t];
    wave_sync();
  };
  auto paint_mean_cov = [&](uint8_t* px, int row) {
    const int mc_val = (row < mc_limit) ? (row < band ? 255 : 200) : -1;
    if (c->mean_cov_channel >= 0 && mc_val >= 0) {
      for (int col = lane; col < W; col += 64) px[col * CO + c->mean_cov_channel] = mc_val;
    }
  };

  // What one read row needs from memory, for one pass of 64 * kColsPerLane columns: the
  // wave-uniform metadata of its read (phase C'), per column the LAST event of the CIGAR walk
  // (ev: 0 nothing, 1 aligned base, 2 indel anchor; ri: index of the read base) and the bytes
  // that event points at.
  struct RowIn {
    int slot;
    uint32_t flags, s0;
    int ev[kColsPerLane], ri[kColsPerLane];
    uint32_t bb[kColsPerLane], qq[kColsPerLane], rr[kColsPerLane];   // zero-extended bytes
  };
  // One CIGAR walk per pass (one pass for W <= 256): the ops come in with one coalesced vector
  // load -- from the LDS cache of phase C' for reads of up to kCigCache operations, else from
  // global memory -- and are broadcast with v_readlane, the running (ref_i, read_i) are
  // wave-uniform (SGPRs), and every lane resolves the LAST event on each of its kColsPerLane
  // columns (lane, lane+64, ...) in registers -- the reference overwrites in op order.  Then the
  // bases / qualities / reference bases of all columns are REQUESTED; nothing waits for them here.
  auto gather = [&](int row, int cb0, RowIn& in, auto lds_only_tag) {
    constexpr bool kLdsOnly = decltype(lds_only_tag)::value;   // every kept read's CIGAR sits in the LDS cache
    const uint8_t* const bases_p = a.bases;
    const uint8_t* const quals_p = a.quals;
    const uint8_t* const ref_p = ref;
    const int slot = __builtin_amdgcn_readfirstlane(order[row - band]);
    in.slot = slot;
    in.flags = __builtin_amdgcn_readfirstlane(m_flags[slot]);
    const uint32_t c0 = __builtin_amdgcn_readfirstlane(m_c0[slot]);
    const uint32_t c1 = __builtin_amdgcn_readfirstlane(m_c1[slot]);
    in.s0 = __builtin_amdgcn_readfirstlane(m_s0[slot]);
    const int rpos = __builtin_amdgcn_readfirstlane(m_rpos[slot]);
#pragma unroll
    for (int q = 0; q < kColsPerLane; ++q) {
      in.ev[q] = 0;
      in.ri[q] = 0;
    }
    int ref_i = rpos, read_i = 0;
    const int n_ops = static_cast<int>(c1 - c0);
    for (int kb = 0; kb < n_ops; kb += 64) {
      uint32_t cg_v;
      if (kLdsOnly || n_ops <= cig_cache) {   // wave-uniform
        cg_v = m_cig[slot * cig_cache + (lane & (cig_cache - 1))];
      } else {
        cg_v = (kb + lane < n_ops) ? a.cigar[c0 + kb + lane] : 0u;
      }
      const int nk = min(64, n_ops - kb);
      for (int k = 0; k < nk; ++k) {
        const uint32_t cg = __builtin_amdgcn_readlane(cg_v, k);
        const int op = cg & 0xF;
        const int len = cg >> 4;
        // Branch-free: every operation writes at most ONE run of events -- columns [ev_start, ev_start + ev_count)
        // receive (ev_code, ri_base + offset in the run):
        //   M / = / X   the aligned bases:              [ref_i, ref_i + len)   code 1, read_i + d
        //   I           its anchor (if ref_i > 0):      [ref_i - 1, ref_i)     code 2, read_i
        //   D           its anchor (if read_i > 0):     [ref_i - 1, ref_i)     code 2, read_i - 1
        //   N, S, H, P  nothing (count 0)
        // All of that is wave-uniform scalar arithmetic; a lane spends five vector instructions per column and
        // operation, without the switch's branches and register copies (a long-read CIGAR has a dozen operations
        // per read).  Same events in the same order as the reference's walk (pileup_channel_lib.cc:171-261).
        const bool is_m = op == DV_CIGAR_ALIGNMENT_MATCH || op == DV_CIGAR_SEQUENCE_MATCH ||
                          op == DV_CIGAR_SEQUENCE_MISMATCH;
        const bool is_i = op == DV_CIGAR_INSERT, is_d = op == DV_CIGAR_DELETE;
        const int ev_start = is_m ? ref_i : ref_i - 1;
        const unsigned ev_count = is_m ? static_cast<unsigned>(len)
                                       : ((is_i && ref_i > 0) || (is_d && read_i > 0)) ? 1u : 0u;
        const int ev_code = is_m ? 1 : 2;
        const int ri_base = is_d ? read_i - 1 : read_i;
#pragma unroll
        for (int q = 0; q < kColsPerLane; ++q) {
          const unsigned d = static_cast<unsigned>(istart + cb0 + q * 64 + lane - ev_start);
          const bool hit = d < ev_count;
          in.ev[q] = hit ? ev_code : in.ev[q];
          in.ri[q] = hit ? ri_base + static_cast<int>(d) : in.ri[q];
        }
        ref_i += (is_m || is_d || op == DV_CIGAR_SKIP) ? len : 0;
        read_i += (is_m || is_i || op == DV_CIGAR_CLIP_SOFT) ? len : 0;
      }
    }
    // UNCONDITIONAL loads (columns without an event read the read's first base / the window's
    // last column and are discarded in draw): behind a per-element condition hipcc branches
    // around every load and waits for it on the spot -- twelve serial round trips per row.
    // They are issued from inline assembly: the compiler's own wait insertion, which cannot
    // count through draw's branches, would otherwise put `s_waitcnt vmcnt(0)` in front of the
    // first use and so wait for the NEXT row's loads as well; wait_row() below waits exactly.
#pragma unroll
    for (int q = 0; q < kColsPerLane; ++q) {
      const int col = cb0 + q * 64 + lane;
      const bool on = col < W && in.ev[q] != 0;
      if (!on) {
        in.ev[q] = 0;
        in.ri[q] = 0;
      }
      const uint32_t at = in.s0 + static_cast<uint32_t>(in.ri[q]);
      const uint32_t rc = static_cast<uint32_t>(min(col, W - 1));
      // `s_nop 4`: the base pointers may have been restored from spill lanes (v_readlane) right
      // in front of the statement, and hipcc pads no hazards for the inside of an asm string
      // (VALU-written SGPR -> VMEM address: 5 wait states)
      asm volatile("s_nop 4\n\tglobal_load_ubyte %0, %1, %2" : "=v"(in.bb[q]) : "v"(at), "s"(bases_p) : "memory");
      asm volatile("s_nop 4\n\tglobal_load_ubyte %0, %1, %2" : "=v"(in.qq[q]) : "v"(at), "s"(quals_p) : "memory");
      asm volatile("s_nop 4\n\tglobal_load_ubyte %0, %1, %2" : "=v"(in.rr[q]) : "v"(rc), "s"(ref_p) : "memory");
    }
  };
  // The 3 * kColsPerLane loads of `in` have landed; the as many loads issued AFTER them (the
  // next row's gather) may still be in flight.  Every destination register passes through the
  // statement, so no use can be scheduled ahead of it.
  static_assert(kColsPerLane == 4, "wait_row lists twelve registers");
  auto wait_row = [&](RowIn& in, bool newer_in_flight) {
    if (newer_in_flight) {
      asm volatile("s_waitcnt vmcnt(12)"
                   : "+v"(in.bb[0]), "+v"(in.bb[1]), "+v"(in.bb[2]), "+v"(in.bb[3]), "+v"(in.qq[0]), "+v"(in.qq[1]),
                     "+v"(in.qq[2]), "+v"(in.qq[3]), "+v"(in.rr[0]), "+v"(in.rr[1]), "+v"(in.rr[2]), "+v"(in.rr[3])
                   :: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(in.bb[0]), "+v"(in.bb[1]), "+v"(in.bb[2]), "+v"(in.bb[3]), "+v"(in.qq[0]), "+v"(in.qq[1]),
                     "+v"(in.qq[2]), "+v"(in.qq[3]), "+v"(in.rr[0]), "+v"(in.rr[1]), "+v"(in.rr[2]), "+v"(in.rr[3])
                   :: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // The pixels of one pass from the gathered bytes, into the LDS row.
  auto draw = [&](const RowIn& in, int cb0, uint8_t* px) {
    const int slot = in.slot;
    const uint32_t flags = in.flags, s0 = in.s0;
    uint32_t konst[kPixDw], sel_a[kPixDw], sel_b[kPixDw];
    bool any_b = false;
#pragma unroll
    for (int d = 0; d < kPixDw; ++d) {
      konst[d] = 0;
      sel_a[d] = sel_b[d] = 0x03020100u;
      if (d < pdw) {
        konst[d] = __builtin_amdgcn_readfirstlane(m_const[slot * pdw + d]);
        sel_a[d] = __builtin_amdgcn_readfirstlane(m_sel_a[slot * pdw + d]);
        sel_b[d] = __builtin_amdgcn_readfirstlane(m_sel_b[slot * pdw + d]);
      }
      any_b |= sel_b[d] != 0x03020100u;
    }
#pragma unroll
    for (int q = 0; q < kColsPerLane; ++q) {
      const int col = cb0 + q * 64 + lane;
      if (col >= W) continue;
      // no event: nothing is drawn; an indel anchor draws the anchor character with the quality
      // of the read base the event points at
      const uint8_t base = in.ev[q] == 0 ? 0 : in.ev[q] == 1 ? static_cast<uint8_t>(in.bb[q])
                                                              : static_cast<uint8_t>(c->anchor_char);
      uint32_t o[kPixDw];
#pragma unroll
      for (int d = 0; d < kPixDw; ++d) o[d] = 0;
      if (base != 0) {  // `read_base &&` (pileup_channel_lib.cc:139): a NUL base draws nothing
        uint32_t dyn_a = c->lut_base[base] | (static_cast<uint32_t>(c->lut_bq[in.qq[q]]) << 8) |
                         (static_cast<uint32_t>(c->diff[base == static_cast<uint8_t>(in.rr[q]) ? 1 : 0]) << 16);
        if ((flags & DV_READ_HAS_5MC) && a.mod_5mc) {
          dyn_a |= static_cast<uint32_t>(c->lut_mod[a.mod_5mc[s0 + in.ri[q]]]) << 24;
        }
#pragma unroll
        for (int d = 0; d < kPixDw; ++d) {
          if (d < n_pix_dw) o[d] = __builtin_amdgcn_perm(dyn_a, konst[d], sel_a[d]);
        }
        if (any_b) {
          uint32_t dyn_b = 0;  // {6mA, is_homopolymer, homopolymer_weighted} pixels of this base
          if ((flags & DV_READ_HAS_6MA) && a.mod_6ma) dyn_b = c->lut_mod[a.mod_6ma[s0 + in.ri[q]]];
          if (a.base_aux0) dyn_b |= static_cast<uint32_t>(a.base_aux0[s0 + in.ri[q]]) << 8;
          if (a.base_aux1) dyn_b |= static_cast<uint32_t>(a.base_aux1[s0 + in.ri[q]]) << 16;
          if (a.base_aux2) dyn_b |= static_cast<uint32_t>(a.base_aux2[s0 + in.ri[q]]) << 24;
#pragma unroll
          for (int d = 0; d < kPixDw; ++d) {
            if (d < n_pix_dw) o[d] = __builtin_amdgcn_perm(dyn_b, o[d], sel_b[d]);
          }
        }
      }
      // every lane writes its whole pixel (zeros where the read draws nothing), so the
      // row buffer needs no clearing pass
      store_pixel(px + col * CO, o, CO);
    }
  };

  // ---- reference rows: EncodeReference (pileup_channel_lib.cc:263-293)
  for (int row = wave; row < min(band, n_render); row += kWaves) {
    uint8_t* px = row_begin(row, false);
    // pixel = per-channel constants, with the base-colour byte spliced in by v_perm
    uint32_t rk[kPixDw], rs[kPixDw];
#pragma unroll
    for (int d = 0; d < kPixDw; ++d) {
      rk[d] = __builtin_amdgcn_readfirstlane(c->ref_konst[d]);
      rs[d] = __builtin_amdgcn_readfirstlane(c->ref_sel[d]);
    }
    const size_t ref_row = static_cast<size_t>(a.item_ref_idx[item]) * W;
    for (int col = lane; col < W; col += 64) {
      uint32_t bv = c->lut_base[ref[col]];
      if (a.ref_aux0) bv |= static_cast<uint32_t>(a.ref_aux0[ref_row + col]) << 8;
      if (a.ref_aux1) bv |= static_cast<uint32_t>(a.ref_aux1[ref_row + col]) << 16;
      if (a.ref_aux2) bv |= static_cast<uint32_t>(a.ref_aux2[ref_row + col]) << 24;
      uint32_t o[kPixDw];
#pragma unroll
      for (int d = 0; d < kPixDw; ++d) o[d] = __builtin_amdgcn_perm(bv, rk[d], rs[d]);
      store_pixel(px + col * CO, o, CO);
    }
    paint_mean_cov(px, row);
    row_flush(row);
  }

  // ---- read rows.  A row's pixels hang on a chain of dependent loads (row order -> CIGAR ->
  // bases): ~1.5 us of latency against ~0.3 us of work, which is what bounded this kernel at
  // 0.28 of the HBM roofline (HISTORY.md 4.1).  The wave therefore runs its rows as a two-stage
  // software pipeline: the NEXT row's walk runs and its byte loads are issued before the
  // CURRENT row's pixels are drawn, so those loads travel under a whole row of work; the CIGAR
  // itself comes from LDS (phase C').  Single-pass images (W <= 256: every BASELINE shape) whose
  // kept reads all have at most kCigCache CIGAR operations; other items take the plain
  // row-at-a-time path.
  const int read_end = band + kept;
  int first_read = band + ((wave - band) % kWaves + kWaves) % kWaves;
  if (W <= 64 * kColsPerLane && !any_long_cigar) {
    if (first_read < read_end) {
      // two named register sets, alternating (a copy `cur = next` would have to wait for the loads
      // it copies): even trips draw `even` while `odd` is gathered, odd trips the other way round.
      // Gathers are unconditional (past the end the wave gathers its last row again): a load
      // count that depends on the path would make the compiler wait for everything at the next use.
      RowIn even, odd;
      const int last = first_read + (read_end - 1 - first_read) / kWaves * kWaves;   // this wave's last read row
      gather(first_read, 0, even, std::true_type{});
      for (int row = first_read; row < read_end; row += 2 * kWaves) {
        gather(min(row + kWaves, last), 0, odd, std::true_type{});
        {
          wait_row(even, true);
          uint8_t* px = row_begin(row, false);
          draw(even, 0, px);
          paint_mean_cov(px, row);
          row_flush(row);
        }
        gather(min(row + 2 * kWaves, last), 0, even, std::true_type{});
        if (row + kWaves < read_end) {
          wait_row(odd, true);
          uint8_t* px = row_begin(row + kWaves, false);
          draw(odd, 0, px);
          paint_mean_cov(px, row + kWaves);
          row_flush(row + kWaves);
        }
      }
      // The loop leaves one or two redundant gathers in flight (12-24 asm loads the compiler cannot
      // see) into `even` / `odd`.  Those registers are dead to the allocator from here on, so a load
      // that lands late would overwrite whatever it puts there next (a store address, a counter):
      // drain them while both sets are still live.  The asm loads must never outlive their
      // destination registers' live range.  Costs nothing -- the tail has nothing to overlap with.
      wait_row(even, false);
      wait_row(odd, false);
    }
  } else {
    for (int row = first_read; row < read_end; row += kWaves) {
      uint8_t* px = row_begin(row, false);
      for (int cb0 = 0; cb0 < W; cb0 += 64 * kColsPerLane) {
        RowIn in;
        gather(row, cb0, in, std::false_type{});
        wait_row(in, false);
        draw(in, cb0, px);
      }
      paint_mean_cov(px, row);
      row_flush(row);
    }
  }

  // ---- rows below the pileup but inside the mean-coverage paint
  {
    int row = read_end + ((wave - read_end) % kWaves + kWaves) % kWaves;
    for (; row < n_render; row += kWaves) {
      uint8_t* px = row_begin(row, true);
      paint_mean_cov(px, row);
      row_flush(row);
    }
  }

  // ---------------- blank rows: 16-byte zero stores -------------------------
  {
    const uint64_t base = reinterpret_cast<uint64_t>(a.out);
    const uint64_t z0 = base + out0 + static_cast<uint64_t>(n_render) * row_bytes;
    const uint64_t z1 = base + out0 + static_cast<uint64_t>(H) * row_bytes;
    if (z1 > z0) {
      const uint64_t a0 = min((z0 + 15) & ~15ull, z1);
      const uint64_t a1 = max(z1 & ~15ull, a0);
      for (uint64_t b = z0 + tid; b < a0; b += kBlock)
        *reinterpret_cast<uint8_t*>(b) = 0;
      uint4* body = reinterpret_cast<uint4*>(a0);
      const uint64_t nvec = (a1 - a0) >> 4;
      const uint4 zero = make_uint4(0, 0, 0, 0);
      for (uint64_t i = tid; i < nvec; i += kBlock) body[i] = zero;
      for (uint64_t b = a1 + tid; b < z1; b += kBlock)
        *reinterpret_cast<uint8_t*>(b) = 0;
    }
  }
  if (tid == 0 && a.out_rows) a.out_rows[item] = kept;
}

// ------------------------------------------------------------------ host side

// channels/base_quality_channel.cc:59-66 and its clones.
inline uint8_t ScaleColor(int value, float max_val) {
  if (static_cast<float>(value) > max_val) value = max_val;
  return static_cast<int>(254.0f * (static_cast<float>(value) / max_val));
}

}  // namespace

struct dv_encoder {
  int device = 0;
  dv_encoder_options opt{};
  EncConst konst{};
  dv::DeviceBuffer d_konst, d_perm_off, d_perm, d_out, d_rows;
  int perm_dense = 0;                    // pi_n is present for every n <= perm_dense
  std::vector<uint32_t> perm_off_host;   // [n] -> offset into perm_host, kPermAbsent if not built
  std::vector<uint16_t> perm_host;
  std::vector<dv::DeviceBuffer> staging;
};

namespace {

int build_const(const dv_encoder_options& o, EncConst* k) {
  memset(k, 0, sizeof(*k));
  // The reference's odd-width CHECK (pileup_image_native.cc:114) belongs to the
  // options object and is enforced by the host mirror: EncodeRead /
  // EncodeReference take the width from ref_bases and accept any length.
  if (o.width < 1 || o.width > 4096) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "width out of range");
  }
  if (o.n_channels < 1 || o.n_channels > DV_MAX_CHANNELS) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "n_channels out of range");
  }
  if (o.reference_band_height < 0 || o.height <= o.reference_band_height) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT,
                    "height must exceed reference_band_height");
  }
  k->n_channels = o.n_channels;
  k->width = o.width;
  k->band = o.reference_band_height;
  k->min_bq = o.min_base_quality;
  k->min_mapq = o.min_mapping_quality;
  k->anchor_char = o.indel_anchoring_base_char & 0xFF;
  k->sort_by_haplotypes = o.sort_by_haplotypes;
  k->polishing = o.hp_tag_for_assembly_polishing;
  k->sort_by_group = o.sort_by_alt_allele_support;
  k->mean_cov_channel = -1;
  // read_base_channel.cc:56-73
  k->lut_base['A'] = o.base_color_offset_a_and_g + o.base_color_stride * 3;
  k->lut_base['G'] = o.base_color_offset_a_and_g + o.base_color_stride * 2;
  k->lut_base['T'] = o.base_color_offset_t_and_c + o.base_color_stride * 1;
  k->lut_base['C'] = o.base_color_offset_t_and_c + o.base_color_stride * 0;
  for (int q = 0; q < 256; ++q) {
    k->lut_bq[q] = ScaleColor(q, o.base_quality_cap);        // base_quality_channel.cc:44-50
    k->lut_mapq[q] = ScaleColor(q, o.mapping_quality_cap);   // mapping_quality_channel.cc:43-51
    k->lut_mod[q] = ScaleColor(q, 255);                      // base_methylation_channel.cc:88-100
  }
  for (int f = 0; f <= 1000; ++f) {
    // insert_size_channel.cc:79-90
    k->lut_insert[f] = static_cast<uint8_t>(
        static_cast<int>(254.0f * (static_cast<float>(f) / 1000.0f)));
  }
  // read_supports_variant_channel.cc:105-116
  const float alphas[3] = {o.allele_unsupporting_read_alpha,
                           o.allele_supporting_read_alpha,
                           o.other_allele_supporting_read_alpha};
  for (int i = 0; i < 3; ++i) {
    k->lut_support[i] = static_cast<uint8_t>(static_cast<int>(254.0f * alphas[i]));
  }
  k->lut_support[3] = k->lut_support[2];
  k->strand[0] = static_cast<uint8_t>(o.positive_strand_color);  // strand_channel.cc:44-61
  k->strand[1] = static_cast<uint8_t>(o.negative_strand_color);
  // base_differs_from_ref_channel.cc:59-66
  k->diff[0] = static_cast<uint8_t>(
      static_cast<int>(254.0f * o.reference_mismatching_read_alpha));
  k->diff[1] = static_cast<uint8_t>(
      static_cast<int>(254.0f * o.reference_matching_read_alpha));
  // supplementary_alignment_channel.cc:49-59
  k->supp[0] = static_cast<unsigned char>(254.0f * o.allele_unsupporting_read_alpha);
  k->supp[1] = static_cast<unsigned char>(254.0f * o.allele_supporting_read_alpha);

  const uint8_t ref_bq = ScaleColor(o.reference_base_quality, o.base_quality_cap);
  for (int c = 0; c < o.n_channels; ++c) {
    uint8_t kind = kZero, ref = 0;
    switch (o.channels[c]) {
      case DV_CH_READ_BASE: kind = kBase; break;
      case DV_CH_BASE_QUALITY: kind = kBaseQual; ref = ref_bq; break;
      case DV_CH_MAPPING_QUALITY: kind = kMapq; ref = ref_bq; break;  // mapping_quality_channel.cc:53-58
      case DV_CH_STRAND: kind = kStrand; ref = k->strand[0]; break;
      case DV_CH_READ_SUPPORTS_VARIANT: kind = kSupport; ref = k->lut_support[0]; break;
      case DV_CH_BASE_DIFFERS_FROM_REF: kind = kDiff; ref = k->diff[1]; break;
      case DV_CH_HAPLOTYPE_TAG: kind = kHaplotype; ref = 0; break;
      case DV_CH_ALLELE_FREQUENCY: kind = kAuxList; ref = 0; break;
      case DV_CH_READ_MAPPING_PERCENT: kind = kAuxRead0; ref = 254; break;
      case DV_CH_AVG_BASE_QUALITY: kind = kAuxRead1; ref = 254; break;
      case DV_CH_IDENTITY: kind = kAuxRead2; ref = 254; break;
      case DV_CH_GAP_COMPRESSED_IDENTITY: kind = kAuxRead3; ref = 254; break;
      case DV_CH_GC_CONTENT: kind = kAuxRead4; break;
      case DV_CH_IS_HOMOPOLYMER: kind = kBaseAux0; break;
      case DV_CH_HOMOPOLYMER_WEIGHTED: kind = kBaseAux1; break;
      case DV_CH_BLANK: kind = kZero; break;
      case DV_CH_INSERT_SIZE: kind = kInsert; ref = 254; break;
      case DV_CH_MEAN_COVERAGE: kind = kZero; k->mean_cov_channel = c; break;
      case DV_CH_BASE_METHYLATION: kind = kMeth5; break;
      case DV_CH_BASE_6MA: kind = kMeth6; break;
      case DV_CH_SUPPLEMENTARY_ALIGNMENT:
        kind = kSupplementary;
        ref = static_cast<uint8_t>(o.allele_unsupporting_read_alpha);  // sic (:61-65)
        break;
      case DV_CH_ALLELE_SAMPLE_PROBABILITY: kind = kAuxList; ref = 0; break;
      case DV_CH_HOMOPOLYMER_INSERTION_QUALITY:
      case DV_CH_HOMOPOLYMER_DELETION_QUALITY:
      case DV_CH_INTER_HOMOPOLYMER_INSERTION_QUALITY: {
        // per-base pixels from the host; the plane follows dv_base_aux_plane's rule; the reference row is 0
        // (homopolymer_insertion_quality_channel.cc:63-67 and its siblings)
        const int plane = dv_base_aux_plane(o.channels, o.n_channels, c);
        if (plane < 0 || plane > 2) return plane < 0 ? plane : DV_ERR_INVALID_ARGUMENT;
        kind = static_cast<uint8_t>(kBaseAux0 + plane);
        break;
      }
      case DV_CH_READ_SUPPORTS_VARIANT_FUZZY:  // read_supports_variant_fuzzy_channel.cc:115-119
        kind = kAuxList;
        ref = k->lut_support[0];
        break;
      default:
        return dv::fail(DV_ERR_UNSUPPORTED,
                        "channel enum " + std::to_string(o.channels[c]) +
                            " is not drawn by the device encoder");
    }
    k->kind[c] = kind;
    k->ref_const[c] = ref;
  }
  {  // list_aux is ONE byte per (item, read): at most one channel can be drawn from it
    int n_list_aux = 0;
    for (int c = 0; c < o.n_channels; ++c) n_list_aux += k->kind[c] == kAuxList ? 1 : 0;
    if (n_list_aux > 1) {
      return dv::fail(DV_ERR_UNSUPPORTED,
                      "at most one of allele_frequency / read_supports_variant_fuzzy / "
                      "allele_sample_probability per channel set");
    }
  }
  for (int d = 0; d < DV_MAX_CHANNELS / 4; ++d) {
    k->ref_konst[d] = 0;
    k->ref_sel[d] = 0x03020100u;
  }
  for (int c = 0; c < o.n_channels; ++c) {
    const int d = c >> 2, sh = (c & 3) * 8;
    // (a flow-space channel may sit in plane 0 / 1 too: its reference row is the constant 0, not ref_aux0 / ref_aux1)
    const int dyn = k->kind[c] == kBase ? 4 : o.channels[c] == DV_CH_IS_HOMOPOLYMER ? 5
                    : o.channels[c] == DV_CH_HOMOPOLYMER_WEIGHTED ? 6 : k->kind[c] == kAuxRead4 ? 7 : -1;
    if (dyn >= 0) {  // reference-row byte taken from {base colour, ref_aux0, ref_aux1, ref_aux2}
      k->ref_sel[d] = (k->ref_sel[d] & ~(0xFFu << sh)) | (static_cast<uint32_t>(dyn) << sh);
    } else {
      k->ref_konst[d] |= static_cast<uint32_t>(k->ref_const[c]) << sh;
    }
  }
  return DV_OK;
}

// pi_n (DownsampleReadIndices' permutation of n read indices) is a pure function of
// (n, seed) because the reference passes the generator by value
// (pileup_image_native.cc:153-165,327,343), so it is a table lookup on the device.
// The table is DENSE for n <= kPermDense (every depth a whole-genome run meets; grown by
// doubling) and SPARSE above it: a pile-up deeper than that (amplicon data) adds only the
// permutations of the depths that actually occur, so a 10,000-deep item costs 20 KB and
// one shuffle instead of a quadratic table of every n below it.
constexpr int kPermDense = 1024;
constexpr uint32_t kPermAbsent = 0xFFFFFFFFu;

void append_perm(dv_encoder* enc, int n) {
  std::vector<int> idx(n);
  std::iota(idx.begin(), idx.end(), 0);
  std::mt19937_64 gen(enc->opt.random_seed);
  std::shuffle(idx.begin(), idx.end(), gen);
  enc->perm_off_host[n] = static_cast<uint32_t>(enc->perm_host.size());
  for (int i = 0; i < n; ++i) enc->perm_host.push_back(static_cast<uint16_t>(idx[i]));
}

// Makes pi_n available for every n <= min(max_n, kPermDense) and for every n in `deep`.
int ensure_perm_table(dv_encoder* enc, int max_n, const std::vector<int>& deep) {
  if (max_n > 65535) {
    return dv::fail(DV_ERR_UNSUPPORTED, "more than 65535 reads in one pileup");
  }
  bool changed = false;
  if (static_cast<int>(enc->perm_off_host.size()) < max_n + 2) {
    enc->perm_off_host.resize(max_n + 2, kPermAbsent);
    changed = true;
  }
  const int want_dense = std::min(max_n, kPermDense);
  if (want_dense > enc->perm_dense) {
    int cap = std::max(256, enc->perm_dense);
    while (cap < want_dense) cap *= 2;
    cap = std::min(cap, kPermDense);
    if (static_cast<int>(enc->perm_off_host.size()) < cap + 2) {
      enc->perm_off_host.resize(cap + 2, kPermAbsent);
    }
    for (int n = enc->perm_dense + 1; n <= cap; ++n) {
      if (enc->perm_off_host[n] == kPermAbsent) append_perm(enc, n);
    }
    enc->perm_dense = cap;
    changed = true;
  }
  for (int n : deep) {
    if (n > enc->perm_dense && enc->perm_off_host[n] == kPermAbsent) {
      append_perm(enc, n);
      changed = true;
    }
  }
  if (!changed) return DV_OK;
  // (reserve() may free the old allocation: hipFree waits for the kernels that read it)
  if (int rc = enc->d_perm_off.reserve(enc->perm_off_host.size() * 4)) return rc;
  if (int rc = enc->d_perm.reserve(std::max<size_t>(enc->perm_host.size() * 2, 16))) return rc;
  DV_HIP_CHECK(hipMemcpy(enc->d_perm_off.ptr, enc->perm_off_host.data(),
                         enc->perm_off_host.size() * 4, hipMemcpyHostToDevice));
  if (!enc->perm_host.empty()) {
    DV_HIP_CHECK(hipMemcpy(enc->d_perm.ptr, enc->perm_host.data(), enc->perm_host.size() * 2,
                           hipMemcpyHostToDevice));
  }
  return DV_OK;
}

// The list lengths above the dense range, from the batch's item_list_off (read back from
// the device for a DV_MEM_DEVICE batch -- only when max_list_len says such items may exist).
int deep_list_lengths(const dv_encoder* enc, const dv_batch* b, hipStream_t stream,
                      std::vector<int>* deep) {
  deep->clear();
  if (static_cast<int>(b->max_list_len) <= kPermDense) return DV_OK;
  std::vector<uint32_t> tmp;
  const uint32_t* off = b->item_list_off;
  if (b->memory != DV_MEM_HOST) {
    tmp.resize(static_cast<size_t>(b->n_items) + 1);
    DV_HIP_CHECK(hipStreamSynchronize(stream));
    DV_HIP_CHECK(hipMemcpy(tmp.data(), b->item_list_off, tmp.size() * 4, hipMemcpyDeviceToHost));
    off = tmp.data();
  }
  for (int i = 0; i < b->n_items; ++i) {
    const uint32_t n = off[i + 1] - off[i];
    if (n > b->max_list_len) return dv::fail(DV_ERR_INVALID_ARGUMENT, "max_list_len too small");
    if (static_cast<int>(n) > kPermDense) deep->push_back(static_cast<int>(n));
  }
  std::sort(deep->begin(), deep->end());
  deep->erase(std::unique(deep->begin(), deep->end()), deep->end());
  (void)enc;
  return DV_OK;
}

template <typename T>
int stage(dv_encoder* enc, size_t slot, const T* host, size_t count,
          hipStream_t stream, const T** dev) {
  if (host == nullptr || count == 0) {
    *dev = nullptr;
    // keep a valid (dummy) pointer for arrays the kernel indexes with 0 items
    if (host != nullptr) {
      if (int rc = enc->staging[slot].reserve(16)) return rc;
      *dev = static_cast<const T*>(enc->staging[slot].ptr);
    }
    return DV_OK;
  }
  if (int rc = enc->staging[slot].reserve(count * sizeof(T))) return rc;
  DV_HIP_CHECK(hipMemcpyAsync(enc->staging[slot].ptr, host, count * sizeof(T),
                              hipMemcpyHostToDevice, stream));
  *dev = static_cast<const T*>(enc->staging[slot].ptr);
  return DV_OK;
}

}  // namespace

extern "C" {

int dv_base_aux_plane(const int32_t* channels, int32_t n_channels, int32_t channel_index) {
  if (!channels || channel_index < 0 || channel_index >= n_channels) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_base_aux_plane: bad channel index");
  }
  bool used[3] = {false, false, false};
  for (int c = 0; c < n_channels; ++c) {
    if (channels[c] == DV_CH_IS_HOMOPOLYMER) used[0] = true;
    if (channels[c] == DV_CH_HOMOPOLYMER_WEIGHTED) used[1] = true;
  }
  if (channels[channel_index] == DV_CH_IS_HOMOPOLYMER) return 0;
  if (channels[channel_index] == DV_CH_HOMOPOLYMER_WEIGHTED) return 1;
  for (int c = 0; c <= channel_index; ++c) {
    const int ch = channels[c];
    if (ch != DV_CH_HOMOPOLYMER_INSERTION_QUALITY && ch != DV_CH_HOMOPOLYMER_DELETION_QUALITY &&
        ch != DV_CH_INTER_HOMOPOLYMER_INSERTION_QUALITY) {
      continue;
    }
    int plane = 2;
    while (plane >= 0 && used[plane]) --plane;
    if (plane < 0) {
      return dv::fail(DV_ERR_UNSUPPORTED,
                      "more than three per-base host-computed channels (is_homopolymer, homopolymer_weighted and "
                      "the flow-space channels) in one channel list");
    }
    used[plane] = true;
    if (c == channel_index) return plane;
  }
  return DV_BASE_AUX_NONE;
}

int dv_encoder_create(const dv_encoder_options* options, int device,
                      dv_encoder** out) {
  if (!options || !out) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_encoder_create: null argument");
  }
  EncConst k;
  if (int rc = build_const(*options, &k)) return rc;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
    return dv::fail(DV_ERR_NO_DEVICE,
                    "no HIP device: libdvhip has no CPU fallback");
  }
  if (device < 0 || device >= n_dev) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "bad device ordinal");
  }
  DV_HIP_CHECK(hipSetDevice(device));
  auto* enc = new dv_encoder();
  enc->device = device;
  enc->opt = *options;
  enc->konst = k;
  enc->staging.resize(40);
  int rc = enc->d_konst.reserve(sizeof(EncConst));
  if (rc == DV_OK) {
    hipError_t e = hipMemcpy(enc->d_konst.ptr, &k, sizeof(k), hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = dv::fail(DV_ERR_HIP, hipGetErrorString(e));
  }
  if (rc == DV_OK) rc = ensure_perm_table(enc, 256, {});
  if (rc != DV_OK) {
    dv_encoder_destroy(enc);
    return rc;
  }
  *out = enc;
  return DV_OK;
}

void dv_encoder_destroy(dv_encoder* enc) {
  if (!enc) return;
  (void)hipSetDevice(enc->device);
  enc->d_konst.release();
  enc->d_perm_off.release();
  enc->d_perm.release();
  enc->d_out.release();
  enc->d_rows.release();
  for (auto& b : enc->staging) b.release();
  delete enc;
}

// What the reference would LOG(FATAL)/CHECK on, plus the index ranges the kernel trusts.
// Host-resident batches only (dv_encode_batch runs it on them itself); a caller that
// uploads its own DV_MEM_DEVICE batch runs it on the host image before the upload.
int dv_validate_batch(const dv_batch* b, int32_t reference_band_height) {
  if (!b) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_validate_batch: null");
  if (b->memory != DV_MEM_HOST) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_validate_batch needs a host-resident batch");
  }
  if (b->n_items < 0 || b->n_reads < 0) return dv::fail(DV_ERR_INVALID_ARGUMENT, "negative counts");
  for (uint32_t i = 0; i < b->n_cigar; ++i) {
    const uint32_t op = b->cigar[i] & 0xF;
    if (op < 1 || op > 9) {
      return dv::fail(DV_ERR_BAD_INPUT, "Unrecognized CIGAR op");  // pileup_channel_lib.cc:252
    }
  }
  for (int i = 0; i < b->n_items; ++i) {
    const int h = b->item_height[i];
    if (h <= reference_band_height || h - reference_band_height > kMaxKept) {
      return dv::fail(DV_ERR_INVALID_ARGUMENT,
                      "item_height must be in (reference_band_height, "
                      "reference_band_height + 256]");
    }
    if (b->item_list_off[i + 1] < b->item_list_off[i] || b->item_list_off[i + 1] > b->n_list) {
      return dv::fail(DV_ERR_INVALID_ARGUMENT, "item_list_off is not a prefix sum within n_list");
    }
    if (b->item_list_off[i + 1] - b->item_list_off[i] > b->max_list_len) {
      return dv::fail(DV_ERR_INVALID_ARGUMENT, "max_list_len too small");
    }
    if (b->item_ref_idx[i] >= b->n_ref_windows) {
      return dv::fail(DV_ERR_INVALID_ARGUMENT, "item_ref_idx out of range");
    }
  }
  for (uint32_t i = 0; i < b->n_list; ++i) {
    if (b->list_read[i] >= static_cast<uint32_t>(b->n_reads)) {
      return dv::fail(DV_ERR_INVALID_ARGUMENT, "list_read out of range");
    }
  }
  for (int r = 0; r < b->n_reads; ++r) {
    if (b->read_cigar_off[r + 1] < b->read_cigar_off[r] || b->read_cigar_off[r + 1] > b->n_cigar ||
        b->read_seq_off[r + 1] < b->read_seq_off[r] || b->read_seq_off[r + 1] > b->n_bases) {
      return dv::fail(DV_ERR_INVALID_ARGUMENT, "read offsets are not prefix sums within the arrays");
    }
    // the cigar's query length must not exceed the stored sequence
    uint64_t qlen = 0;
    for (uint32_t c = b->read_cigar_off[r]; c < b->read_cigar_off[r + 1]; ++c) {
      const uint32_t op = b->cigar[c] & 0xF;
      if (op == 1 || op == 2 || op == 5 || op == 8 || op == 9) qlen += b->cigar[c] >> 4;
    }
    if (qlen > b->read_seq_off[r + 1] - b->read_seq_off[r]) {
      return dv::fail(DV_ERR_BAD_INPUT, "CIGAR consumes more bases than aligned_sequence has");
    }
  }
  return DV_OK;
}

int dv_encode_batch(dv_encoder* enc, const dv_batch* b, int out_channels,
                    uint8_t* out, int32_t* out_rows, int out_memory,
                    void* stream_v) {
  if (!enc || !b || !out) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_encode_batch: null argument");
  }
  if (out_channels < enc->opt.n_channels || out_channels > 64) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT,
                    "out_channels must be >= n_channels");
  }
  if (b->n_items == 0) return DV_OK;
  if (b->n_items < 0 || b->n_reads < 0) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "negative counts");
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  DV_HIP_CHECK(hipSetDevice(enc->device));
  {
    std::vector<int> deep;
    if (int rc = deep_list_lengths(enc, b, stream, &deep)) return rc;
    if (int rc = ensure_perm_table(enc, static_cast<int>(b->max_list_len), deep)) return rc;
  }

  const int W = enc->opt.width;
  const size_t row_bytes = static_cast<size_t>(W) * out_channels;
  EncArgs a{};
  a.konst = static_cast<const EncConst*>(enc->d_konst.ptr);
  a.perm_off = static_cast<const uint32_t*>(enc->d_perm_off.ptr);
  a.perm = static_cast<const uint16_t*>(enc->d_perm.ptr);
  a.n_items = b->n_items;
  a.out_channels = out_channels;
  a.n_channels = enc->konst.n_channels;
  a.row_buf_bytes = static_cast<int>((row_bytes + 16 + 15) & ~size_t(15));

  size_t out_bytes = 0, covered_bytes = 0;
  if (b->memory == DV_MEM_HOST) {
    // Validate what the reference would LOG(FATAL)/CHECK on, then stage.
    if (int rc = dv_validate_batch(b, enc->opt.reference_band_height)) return rc;
    for (int i = 0; i < b->n_items; ++i) {
      out_bytes = std::max<size_t>(out_bytes, b->item_out_off[i] + b->item_height[i] * row_bytes);
      covered_bytes += b->item_height[i] * row_bytes;
    }
    size_t s = 0;
    const size_t nr = b->n_reads, ni = b->n_items;
#define DV_STAGE(field, count)                                               \
  if (int rc = stage(enc, s++, b->field, count, stream, &a.field)) return rc;
    DV_STAGE(read_pos, nr)
    DV_STAGE(read_sort_pos, nr)
    DV_STAGE(read_seq_off, nr + 1)
    DV_STAGE(read_cigar_off, nr + 1)
    DV_STAGE(read_mapq, nr)
    DV_STAGE(read_flags, nr)
    DV_STAGE(read_frag_len, nr)
    DV_STAGE(read_hp, nr)
    DV_STAGE(read_name_rank, nr)
    DV_STAGE(read_aux, nr * DV_READ_AUX_STRIDE)
    DV_STAGE(bases, b->n_bases)
    DV_STAGE(quals, b->n_bases)
    DV_STAGE(mod_5mc, b->n_bases)
    DV_STAGE(mod_6ma, b->n_bases)
    DV_STAGE(cigar, b->n_cigar)
    DV_STAGE(item_variant_start, ni)
    DV_STAGE(item_image_start, ni)
    DV_STAGE(item_ref_idx, ni)
    DV_STAGE(item_list_off, ni + 1)
    DV_STAGE(item_height, ni)
    DV_STAGE(item_out_off, ni)
    DV_STAGE(item_blank_mask, ni)
    DV_STAGE(item_mean_coverage, ni)
    DV_STAGE(ref_windows, static_cast<size_t>(b->n_ref_windows) * W)
    DV_STAGE(list_read, b->n_list)
    DV_STAGE(list_code, b->n_list)
    DV_STAGE(list_group, b->n_list)
    DV_STAGE(list_aux, b->n_list)
    DV_STAGE(base_aux0, b->n_bases)
    DV_STAGE(base_aux1, b->n_bases)
    DV_STAGE(base_aux2, b->n_bases)
    DV_STAGE(ref_aux0, static_cast<size_t>(b->n_ref_windows) * W)
    DV_STAGE(ref_aux1, static_cast<size_t>(b->n_ref_windows) * W)
    DV_STAGE(ref_aux2, static_cast<size_t>(b->n_ref_windows) * W)
#undef DV_STAGE
  } else {
#define DV_PASS(field) a.field = b->field;
    DV_PASS(read_pos) DV_PASS(read_sort_pos) DV_PASS(read_seq_off)
    DV_PASS(read_cigar_off) DV_PASS(read_mapq) DV_PASS(read_flags)
    DV_PASS(read_frag_len) DV_PASS(read_hp) DV_PASS(read_name_rank)
    DV_PASS(read_aux) DV_PASS(bases) DV_PASS(quals) DV_PASS(mod_5mc)
    DV_PASS(mod_6ma) DV_PASS(cigar) DV_PASS(item_variant_start)
    DV_PASS(item_image_start) DV_PASS(item_ref_idx) DV_PASS(item_list_off)
    DV_PASS(item_height) DV_PASS(item_out_off) DV_PASS(item_blank_mask)
    DV_PASS(item_mean_coverage) DV_PASS(ref_windows) DV_PASS(list_read)
    DV_PASS(list_code) DV_PASS(list_group) DV_PASS(list_aux)
    DV_PASS(base_aux0) DV_PASS(base_aux1) DV_PASS(base_aux2) DV_PASS(ref_aux0) DV_PASS(ref_aux1) DV_PASS(ref_aux2)
#undef DV_PASS
  }

  uint8_t* d_out = out;
  int32_t* d_rows = out_rows;
  if (out_memory == DV_MEM_HOST) {
    if (b->memory != DV_MEM_HOST) {
      return dv::fail(DV_ERR_INVALID_ARGUMENT,
                      "host output requires a host batch (sizes unknown)");
    }
    if (int rc = enc->d_out.reserve(out_bytes)) return rc;
    d_out = static_cast<uint8_t*>(enc->d_out.ptr);
    // The staging buffer is reused between calls and comes back whole: where the items leave
    // gaps (the row blocks of alt images a candidate does not have) it must read as zero, not
    // as the previous call's pixels.
    if (covered_bytes < out_bytes) DV_HIP_CHECK(hipMemsetAsync(d_out, 0, out_bytes, stream));
    if (out_rows) {
      if (int rc = enc->d_rows.reserve(sizeof(int32_t) * b->n_items)) return rc;
      d_rows = static_cast<int32_t*>(enc->d_rows.ptr);
    }
  }
  if ((reinterpret_cast<uintptr_t>(d_out) & 3) != 0) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "out must be 4-byte aligned");
  }
  a.out = d_out;
  a.out_rows = d_rows;

  // CIGAR cache geometry (EncArgs::cig_cache, kept_cap): from the batch itself when it is host memory, from its
  // ABI v7 hints when it is not; without either the round-4 shape (8 words x kMaxKept reads).  The cache may
  // take up to 24 KB (64 words x 96 reads: the ONT shape), which still leaves three workgroups per CU.
  {
    uint32_t ops = b->max_cigar_ops, height = b->max_item_height;
    if (b->memory == DV_MEM_HOST) {
      ops = 0;
      height = 0;
      for (int r = 0; r < b->n_reads; ++r) ops = std::max(ops, b->read_cigar_off[r + 1] - b->read_cigar_off[r]);
      for (int i = 0; i < b->n_items; ++i) height = std::max<uint32_t>(height, b->item_height[i]);
    }
    const int band = enc->opt.reference_band_height;
    a.kept_cap = height > static_cast<uint32_t>(band) ? std::min<int>(kMaxKept, static_cast<int>(height) - band) : kMaxKept;
    a.kept_cap = (a.kept_cap + 3) & ~3;
    int cache = kCigCache;
    while (cache < kCigCacheMax && static_cast<uint32_t>(cache) < ops) cache *= 2;
    while (cache > kCigCache && static_cast<size_t>(cache) * a.kept_cap * 4 > 24 * 1024) cache /= 2;
    a.cig_cache = ops == 0 ? kCigCache : cache;
    static const int force_cache = getenv("DV_CIG_CACHE") ? atoi(getenv("DV_CIG_CACHE")) : 0;   // tuning knob: 8..64
    if (force_cache >= kCigCache && force_cache <= kCigCacheHardMax && (force_cache & (force_cache - 1)) == 0) {
      a.cig_cache = force_cache;
      if (getenv("DV_CIG_KEPT_MAX") != nullptr) a.kept_cap = kMaxKept;
    }
  }
  const size_t lds = sizeof(EncConst) + 6 * kMaxKept * 4 + 8 * 4 +
                     3 * static_cast<size_t>(a.kept_cap) * ((a.n_channels + 3) / 4) * 4 +
                     static_cast<size_t>(a.kept_cap) * a.cig_cache * 4 +
                     static_cast<size_t>(kWaves) * a.row_buf_bytes;
  {
    dv::ProfileScope prof(dv::kProfEncoder, stream);
    hipLaunchKernelGGL(encode_items_kernel, dim3(b->n_items), dim3(kBlock), lds,
                       stream, a);
  }
  DV_HIP_CHECK(hipGetLastError());
  if (out_memory == DV_MEM_HOST) {
    DV_HIP_CHECK(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, stream));
    if (out_rows) {
      DV_HIP_CHECK(hipMemcpyAsync(out_rows, d_rows, sizeof(int32_t) * b->n_items,
                                  hipMemcpyDeviceToHost, stream));
    }
    DV_HIP_CHECK(hipStreamSynchronize(stream));
  }
  return DV_OK;
}

}  // extern "C"
