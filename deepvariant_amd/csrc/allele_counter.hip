// allele_counter.hip -- AlleleCounter::Add for a whole region's reads in one launch
// (the per-read, per-base integer work in front of the hot path).
//
// Replaces the read loop around AlleleCounter::Add (deepvariant/allelecounter.cc:873-979,
// called per read from make_examples_core.py's region processor) together with
// MakeIndelReadAllele (:402-469), GetPrevBase (:386-400), CanBasesBeUsed (:206-229) and
// AddReadAlleles (:471-543).  Input is the packed read table the encoder already uses
// (dv_batch's read fields); output is what AlleleCount holds per position:
//   * ref_supporting_read_count[interval length]          (atomic adds)
//   * one EVENT per non-reference read allele that landed in the interval
//     (position, read, type, low-quality flag, where its bases are) -- the host turns
//     events into read_alleles maps / allele sums; no string ever exists on the device.
// One wave per read: CIGAR operations are walked in order (wave-uniform), the bases of an
// alignment-match run are handled 64 at a time, indel quality / canonical-base checks are
// wave reductions.  The reference's "an indel supersedes the base it is anchored on" rule
// (AddReadAlleles: of two consecutive read alleles at the same position the first is
// dropped) is a one-entry pending slot per wave.
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <memory>
#include <vector>

#include "dv_internal.h"

// Large results live in pinned host memory (the count array of a 7.7 Mb interval is 31 MB: 1.3 ms
// over PCIe from pinned memory, 5 ms into pageable); small ones -- a 1 kb calling region's 4 KB of
// counts and a few hundred events, once or twice per region -- in ordinary memory: pinning and
// unpinning a buffer costs more than the whole call.
template <typename T>
struct PinnedArray {
  T* ptr = nullptr;
  size_t n = 0;
  bool pinned = false;
  int reserve(size_t count) {
    release();
    if (count == 0) return DV_OK;
    if (count * sizeof(T) < (1u << 20)) {
      ptr = static_cast<T*>(std::malloc(count * sizeof(T)));
      if (!ptr) return dv::fail(DV_ERR_OUT_OF_MEMORY, "malloc");
      pinned = false;
    } else {
      hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&ptr), count * sizeof(T), hipHostMallocDefault);
      if (e != hipSuccess) {
        ptr = nullptr;
        return dv::fail(DV_ERR_OUT_OF_MEMORY, std::string("hipHostMalloc: ") + hipGetErrorString(e));
      }
      pinned = true;
    }
    n = count;
    return DV_OK;
  }
  void release() {
    if (ptr) {
      if (pinned) {
        (void)hipHostFree(ptr);
      } else {
        std::free(ptr);
      }
    }
    ptr = nullptr;
    n = 0;
  }
  ~PinnedArray() { release(); }
};

// Grow-only pinned staging for the batch entry point (one per host thread and direction).
struct PinnedBytes {
  uint8_t* ptr = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return DV_OK;
    if (ptr) (void)hipHostFree(ptr);
    ptr = nullptr;
    cap = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 2, 1u << 20);
    if (hipHostMalloc(reinterpret_cast<void**>(&ptr), want, hipHostMallocDefault) != hipSuccess) {
      ptr = nullptr;
      return dv::fail(DV_ERR_OUT_OF_MEMORY, "hipHostMalloc (allele counter staging)");
    }
    cap = want;
    return DV_OK;
  }
};

struct dv_allele_counts {
  PinnedArray<int32_t> ref_count;
  PinnedArray<dv_allele_event> raw;      // as the kernel left them
  std::vector<dv_allele_event> events;   // sorted by (position, read, read_offset)
  std::vector<int32_t> empty_counts;
  int64_t length = 0;
  int32_t n_reads_counted = 0;
};

namespace {

enum : int { kRef = 1, kSub = 2, kIns = 3, kDel = 4, kSoft = 5 };   // AlleleType
enum : int { opM = 1, opI = 2, opD = 3, opN = 4, opS = 5, opH = 6, opP = 7, opEQ = 8, opX = 9 };

struct CountArgs {
  int32_t n_reads;
  const int32_t* read_pos;
  const uint32_t* seq_off;
  const uint32_t* cigar_off;
  const uint8_t* mapq;
  const uint8_t* bases;
  const uint8_t* quals;
  const uint32_t* cigar;
  const uint8_t* ref;        // reference bases of [ref_start, ref_start + n_ref)
  int64_t ref_start, n_ref;
  int64_t reads_start, reads_end;   // IsValidRefOffset: the reads interval
  int64_t interval_start, interval_len;
  int64_t contig_len;
  int32_t min_mapq, min_bq, legacy;
  int32_t* ref_count;        // [interval_len]
  const uint32_t* candidate_mask;   // track_ref_reads: bit p set = keep the reference reads of position p by name
  dv_allele_event* events;
  uint32_t event_cap;
  uint32_t* counters;        // [0] events wanted, [1] reads counted, [2] reference window too small
};

__device__ __forceinline__ bool canonical(uint8_t b) { return b == 'A' || b == 'C' || b == 'G' || b == 'T'; }

struct Entry {
  bool have;
  int64_t abs_pos;           // absolute reference position of the allele
  int type, low;
  uint32_t read_offset, length;
};

// Reference matches are by far the most frequent outcome (one per aligned base) and pile up
// ~coverage-deep on neighbouring positions: a workgroup counts them in an LDS window that
// starts at its first read and flushes the non-zero counters once, so the device-scope
// atomics shrink by about the pile-up depth of the group's reads; positions outside the
// window (unsorted or very long reads) go straight to memory.
constexpr int kWindow = 4096;        // positions per workgroup window (16 KB of LDS)
constexpr int kReadsPerWave = 16;    // a workgroup of 4 waves takes 64 consecutive reads

constexpr int kBlockEvents = 1024;   // events staged per workgroup before they take global slots

struct BlockState {
  int window[kWindow];
  dv_allele_event events[kBlockEvents];
  int n_events, n_reads;
  uint32_t event_base;
};

__device__ __forceinline__ void emit(const CountArgs& a, uint32_t read, const Entry& e, BlockState* bs,
                                     int64_t window_start) {
  int* window = bs->window;
  const int64_t p = e.abs_pos - a.interval_start;
  if (p < 0 || p >= a.interval_len) return;            // IsValidIntervalOffset
  if (e.type == kRef) {
    if (!e.low) {
      const int64_t w = e.abs_pos - window_start;
      if (w >= 0 && w < kWindow) {
        atomicAdd(&window[w], 1);
      } else {
        atomicAdd(&a.ref_count[p], 1);
      }
    }
    // a REFERENCE read allele exists only where a candidate will be called (allelecounter.cc:504-512)
    if (!a.candidate_mask || !((a.candidate_mask[p >> 5] >> (p & 31)) & 1u)) return;
  }
  dv_allele_event ev;
  ev.position = static_cast<int32_t>(p);
  ev.read = read;
  ev.read_offset = e.read_offset;
  ev.length_type = (static_cast<uint32_t>(e.length) & 0x0fffffffu) | (static_cast<uint32_t>(e.type) << 28) |
                   (e.low ? 0x80000000u : 0u);
  // One counter for the whole launch serialises at ~20 ns per atomic (a quarter of a million
  // events = the whole kernel time): stage in LDS, take the global slots once per workgroup.
  const int local = atomicAdd(&bs->n_events, 1);
  if (local < kBlockEvents) {
    bs->events[local] = ev;
  } else {
    const uint32_t slot = atomicAdd(&a.counters[0], 1u);
    if (slot < a.event_cap) a.events[slot] = ev;
  }
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// One base of an alignment-match run -> its read allele (or none).
__device__ __forceinline__ Entry base_entry(const CountArgs& a, uint32_t s0, int64_t abs_pos, uint32_t b) {
  Entry e{};
  if (abs_pos < a.reads_start || abs_pos >= a.reads_end) return e;        // IsValidRefOffset
  const uint8_t base = a.bases[s0 + b], q = a.quals[s0 + b];
  if (!canonical(base) || (a.legacy && q < a.min_bq)) return e;            // CanBasesBeUsed(len 1)
  e.have = true;
  e.abs_pos = abs_pos;
  e.type = a.ref[abs_pos - a.ref_start] == base ? kRef : kSub;
  e.low = (!a.legacy && q < a.min_bq) ? 1 : 0;
  e.read_offset = b;
  e.length = 1;
  return e;
}

__device__ __forceinline__ void count_read(const CountArgs& a, uint32_t r, int lane, BlockState* window,
                                           int64_t window_start) {
  if (a.mapq[r] < a.min_mapq) return;
  if (lane == 0) atomicAdd(&window->n_reads, 1);
  const uint32_t s0 = a.seq_off[r];
  uint32_t read_offset = 0;
  int64_t abs_pos = a.read_pos[r];
  Entry pending{};
  for (uint32_t c = a.cigar_off[r]; c < a.cigar_off[r + 1]; ++c) {
    const uint32_t word = a.cigar[c];
    const int op = word & 15;
    const uint32_t n = word >> 4;
    if (op == opM || op == opEQ || op == opX) {
      // nothing later can share the pending entry's position any more
      if (pending.have && lane == 0) emit(a, r, pending, window, window_start);
      pending.have = false;
      for (uint32_t i = lane; i + 1 < n; i += 64) {
        const Entry e = base_entry(a, s0, abs_pos + i, read_offset + i);
        if (e.have) emit(a, r, e, window, window_start);
      }
      if (n > 0) pending = base_entry(a, s0, abs_pos + n - 1, read_offset + n - 1);   // may be superseded
      read_offset += n;
      abs_pos += n;
    } else if (op == opI || op == opS || op == opD) {
      // MakeIndelReadAllele: anchored on the base before it.  An allele outside the interval is
      // dropped by AddReadAlleles whatever it is (and so is anything it could supersede), so
      // its validity -- which may need reference bases far from the interval -- is not computed.
      Entry e{};
      int prev = -1;
      const int64_t rel = abs_pos - 1 - a.interval_start;
      const bool wanted = rel >= 0 && rel < a.interval_len;
      if (!wanted) {
        // fall through with prev = -1: no entry
      } else if (read_offset == 0) {
        const int64_t pp = abs_pos - 1;                                  // RefBases(ref_offset - 1, 1)
        if (pp >= 0 && pp < a.contig_len) {
          if (pp >= a.ref_start && pp < a.ref_start + a.n_ref) {
            prev = a.ref[pp - a.ref_start];
          } else if (lane == 0) {
            atomicAdd(&a.counters[2], 1u);
          }
        }
      } else {
        prev = a.bases[s0 + read_offset - 1];
      }
      bool ok = prev >= 0 && canonical(static_cast<uint8_t>(prev));
      int low = 0;
      if (ok && op != opD) {                                             // CanBasesBeUsed over the run
        int qsum = 0, bad = 0;
        for (uint32_t i = lane; i < n; i += 64) {
          const uint8_t b = a.bases[s0 + read_offset + i], q = a.quals[s0 + read_offset + i];
          qsum += q;
          bad += (!canonical(b) || (a.legacy && q < a.min_bq)) ? 1 : 0;
        }
        qsum = wave_sum(qsum);
        bad = wave_sum(bad);
        ok = bad == 0;
        low = (!a.legacy && static_cast<int64_t>(qsum) < static_cast<int64_t>(a.min_bq) * n) ? 1 : 0;
      }
      if (ok && op == opD) {                                             // the deleted reference bases
        if (n == 0 || abs_pos < 0 || abs_pos + n > a.contig_len) {
          ok = false;
        } else if (abs_pos < a.ref_start || abs_pos + n > a.ref_start + a.n_ref) {
          ok = false;
          if (lane == 0) atomicAdd(&a.counters[2], 1u);
        } else {
          int bad = 0;
          for (uint32_t i = lane; i < n; i += 64) bad += canonical(a.ref[abs_pos - a.ref_start + i]) ? 0 : 1;
          ok = wave_sum(bad) == 0;
        }
      }
      if (ok) {
        e.have = true;
        e.abs_pos = abs_pos - 1;
        e.type = op == opD ? kDel : op == opI ? kIns : kSoft;
        e.low = low;
        e.read_offset = read_offset;
        e.length = n;
      }
      // AddReadAlleles: of two consecutive alleles at one position the first is dropped.  A
      // skipped allele (position -1 in the reference) never equals a real position.
      if (pending.have && !(e.have && e.abs_pos == pending.abs_pos) && lane == 0) emit(a, r, pending, window, window_start);
      pending = e;
      if (op == opD) {
        abs_pos += n;
      } else {
        read_offset += n;
      }
    } else if (op == opN || op == opP) {
      abs_pos += n;
    }
  }
  if (pending.have && lane == 0) emit(a, r, pending, window, window_start);
}

__global__ __launch_bounds__(256) void count_alleles_kernel(CountArgs a) {
  __shared__ BlockState bs;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t first = blockIdx.x * (4 * kReadsPerWave);
  for (int i = threadIdx.x; i < kWindow; i += 256) bs.window[i] = 0;
  if (threadIdx.x == 0) bs.n_events = bs.n_reads = 0;
  const int64_t window_start = a.read_pos[first];      // reads arrive position-sorted (BAM order)
  __syncthreads();
  for (int k = 0; k < kReadsPerWave; ++k) {
    const uint32_t r = first + k * 4 + wave;           // neighbouring reads side by side on the four waves
    if (r < static_cast<uint32_t>(a.n_reads)) count_read(a, r, lane, &bs, window_start);
  }
  __syncthreads();
  const int n_ev = bs.n_events < kBlockEvents ? bs.n_events : kBlockEvents;
  if (threadIdx.x == 0) {
    bs.event_base = n_ev ? atomicAdd(&a.counters[0], static_cast<uint32_t>(n_ev)) : 0u;
    if (bs.n_reads) atomicAdd(&a.counters[1], static_cast<uint32_t>(bs.n_reads));
  }
  for (int i = threadIdx.x; i < kWindow; i += 256) {
    const int v = bs.window[i];
    const int64_t p = window_start + i - a.interval_start;
    if (v != 0 && p >= 0 && p < a.interval_len) atomicAdd(&a.ref_count[p], v);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n_ev; i += 256) {
    const uint32_t slot = bs.event_base + i;
    if (slot < a.event_cap) a.events[slot] = bs.events[i];
  }
}

template <typename T>
int to_device(dv::DeviceBuffer& buf, const T* src, size_t count, int memory, const T** out, hipStream_t stream) {
  if (memory != DV_MEM_HOST) {
    *out = src;
    return DV_OK;
  }
  if (int rc = buf.reserve_on_current_device(std::max<size_t>(count, 1) * sizeof(T))) return rc;
  DV_HIP_CHECK(hipMemcpyAsync(buf.ptr, src, count * sizeof(T), hipMemcpyHostToDevice, stream));
  *out = static_cast<const T*>(buf.ptr);
  return DV_OK;
}

// Order: (position, read, read_offset).  One read can leave two alleles at one position when
// a skipped allele sits between them (1I 4S 2D with an unusable soft clip: the insertion is
// not superseded, the deletion is added after it); read offsets order them as the CIGAR
// does, so the consumer's "later entry overwrites" matches read_alleles[key] = allele.
// Large lists: LSD radix sort on (position << 32 | read), 16 bits a pass, then the rare ties by offset.
void order_events(dv_allele_counts* res, size_t n_ev) {
  if (n_ev < (1u << 15)) {
    // a calling region's few hundred events: a comparison sort on (position, read, read_offset);
    // the radix passes below cost 65536-entry histograms each, more than the whole launch
    res->events.assign(res->raw.ptr, res->raw.ptr + n_ev);
    std::stable_sort(res->events.begin(), res->events.end(), [](const dv_allele_event& x, const dv_allele_event& y) {
      if (x.position != y.position) return static_cast<uint32_t>(x.position) < static_cast<uint32_t>(y.position);
      if (x.read != y.read) return x.read < y.read;
      return x.read_offset < y.read_offset;
    });
    res->raw.release();
  } else {
    std::vector<uint64_t> key(n_ev), key2(n_ev);
    std::vector<uint32_t> idx(n_ev), idx2(n_ev);
    uint64_t all = 0;
    for (size_t i = 0; i < n_ev; ++i) {
      key[i] = (static_cast<uint64_t>(static_cast<uint32_t>(res->raw.ptr[i].position)) << 32) | res->raw.ptr[i].read;
      idx[i] = static_cast<uint32_t>(i);
      all |= key[i];
    }
    std::vector<size_t> count(65537);
    for (int shift = 0; shift < 64; shift += 16) {
      if (((all >> shift) & 0xffffu) == 0) continue;      // these 16 bits are zero everywhere
      std::fill(count.begin(), count.end(), size_t{0});
      for (size_t i = 0; i < n_ev; ++i) ++count[((key[i] >> shift) & 0xffffu) + 1];
      for (int c = 0; c < 65536; ++c) count[c + 1] += count[c];
      for (size_t i = 0; i < n_ev; ++i) {
        const size_t d = count[(key[i] >> shift) & 0xffffu]++;
        key2[d] = key[i];
        idx2[d] = idx[i];
      }
      key.swap(key2);
      idx.swap(idx2);
    }
    res->events.resize(n_ev);
    for (size_t i = 0; i < n_ev; ++i) res->events[i] = res->raw.ptr[idx[i]];
    for (size_t i = 0; i + 1 < n_ev;) {
      size_t j = i + 1;
      while (j < n_ev && key[j] == key[i]) ++j;
      if (j - i > 1) {
        std::stable_sort(res->events.begin() + i, res->events.begin() + j,
                         [](const dv_allele_event& x, const dv_allele_event& y) { return x.read_offset < y.read_offset; });
      }
      i = j;
    }
    res->raw.release();
  }
}

// Argument and read-table checks shared by the one-region and the batch entry point.
int check_request(const dv_batch* b, const dv_allele_counter_options* o, dv_allele_counts** out, const char* who) {
  const std::string name(who);
  if (!b || !o || !out) return dv::fail(DV_ERR_INVALID_ARGUMENT, name + ": null");
  if (b->n_reads < 0 || o->interval_end < o->interval_start || !o->ref_bases || o->n_ref_bases <= 0 ||
      o->min_base_quality < 0 || o->min_mapping_quality < 0) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, name + ": bad argument");
  }
  if (b->n_reads && (!b->read_pos || !b->read_seq_off || !b->read_cigar_off || !b->read_mapq || !b->bases ||
                     !b->quals || !b->cigar)) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, name + ": read table field missing");
  }
  if (b->memory == DV_MEM_HOST) {
    // the same read-table checks dv_validate_batch applies (a device-resident table is the
    // caller's to validate before the upload, as for dv_encode_batch)
    for (int32_t r = 0; r < b->n_reads; ++r) {
      if (b->read_cigar_off[r + 1] < b->read_cigar_off[r] || b->read_cigar_off[r + 1] > b->n_cigar ||
          b->read_seq_off[r + 1] < b->read_seq_off[r] || b->read_seq_off[r + 1] > b->n_bases) {
        return dv::fail(DV_ERR_INVALID_ARGUMENT, "read offsets are not prefix sums within the arrays");
      }
      uint64_t qlen = 0;
      for (uint32_t c = b->read_cigar_off[r]; c < b->read_cigar_off[r + 1]; ++c) {
        const uint32_t op = b->cigar[c] & 0xF;
        if (op < 1 || op > 9) return dv::fail(DV_ERR_BAD_INPUT, "Unrecognized CIGAR op");
        if (op == 1 || op == 2 || op == 5 || op == 8 || op == 9) qlen += b->cigar[c] >> 4;
      }
      if (qlen > b->read_seq_off[r + 1] - b->read_seq_off[r]) {
        return dv::fail(DV_ERR_BAD_INPUT, "CIGAR consumes more bases than aligned_sequence has");
      }
    }
  }
  const int64_t reads_start = std::min(o->interval_start, o->reads_interval_start);
  const int64_t reads_end = std::max(o->interval_end, o->reads_interval_end);
  if (o->ref_start > reads_start || o->ref_start + o->n_ref_bases < reads_end) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, name + ": the reference window must cover the reads interval");
  }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
    return dv::fail(DV_ERR_NO_DEVICE, name + ": no HIP device (there is no CPU fallback)");
  }
  return DV_OK;
}

}  // namespace

extern "C" {

int dv_count_alleles(const dv_batch* b, const dv_allele_counter_options* o, dv_allele_counts** out, void* stream_v) {
  if (int rc = check_request(b, o, out, "dv_count_alleles")) return rc;
  const int64_t reads_start = std::min(o->interval_start, o->reads_interval_start);
  const int64_t reads_end = std::max(o->interval_end, o->reads_interval_end);
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  const int64_t len = o->interval_end - o->interval_start;
  auto res = std::make_unique<dv_allele_counts>();
  res->length = len;
  if (b->n_reads == 0 || len == 0) {
    res->empty_counts.assign(static_cast<size_t>(len), 0);
    *out = res.release();
    return DV_OK;
  }
  // result / reference scratch is kept per host thread (grow-only): a region driver calls this
  // once per region and hipMalloc + hipFree of ~100 MB cost more than the kernel
  static thread_local dv::DeviceBuffer d_ref, d_cnt, d_ev, d_ctr, d_mask;
  // ... and so are the upload buffers of a host-resident table: seven hipMalloc + hipFree per call
  // were most of the millisecond a 300-read region took (the kernel itself is a few microseconds)
  static thread_local dv::DeviceBuffer up[7];
  CountArgs a{};
  a.n_reads = b->n_reads;
  const size_t n = static_cast<size_t>(b->n_reads);
  if (int rc = to_device(up[0], b->read_pos, n, b->memory, &a.read_pos, stream)) return rc;
  if (int rc = to_device(up[1], b->read_seq_off, n + 1, b->memory, &a.seq_off, stream)) return rc;
  if (int rc = to_device(up[2], b->read_cigar_off, n + 1, b->memory, &a.cigar_off, stream)) return rc;
  if (int rc = to_device(up[3], b->read_mapq, n, b->memory, &a.mapq, stream)) return rc;
  if (int rc = to_device(up[4], b->bases, b->n_bases, b->memory, &a.bases, stream)) return rc;
  if (int rc = to_device(up[5], b->quals, b->n_bases, b->memory, &a.quals, stream)) return rc;
  if (int rc = to_device(up[6], b->cigar, b->n_cigar, b->memory, &a.cigar, stream)) return rc;
  if (int rc = d_ref.reserve_on_current_device(static_cast<size_t>(o->n_ref_bases))) return rc;
  DV_HIP_CHECK(hipMemcpyAsync(d_ref.ptr, o->ref_bases, static_cast<size_t>(o->n_ref_bases), hipMemcpyHostToDevice,
                              stream));
  a.ref = static_cast<const uint8_t*>(d_ref.ptr);
  a.ref_start = o->ref_start;
  a.n_ref = o->n_ref_bases;
  a.reads_start = reads_start;
  a.reads_end = reads_end;
  a.interval_start = o->interval_start;
  a.interval_len = len;
  a.contig_len = o->contig_n_bases > 0 ? o->contig_n_bases : o->ref_start + o->n_ref_bases;
  a.min_mapq = o->min_mapping_quality;
  a.min_bq = o->min_base_quality;
  a.legacy = o->keep_legacy_behavior ? 1 : 0;
  size_t n_candidate_refs = 0;
  if (o->track_ref_reads && o->n_candidate_positions > 0) {
    if (!o->candidate_positions) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_count_alleles: candidate_positions is null");
    std::vector<uint32_t> mask(static_cast<size_t>((len + 31) / 32), 0u);
    for (int32_t k = 0; k < o->n_candidate_positions; ++k) {
      const int64_t p = o->candidate_positions[k] - o->interval_start;
      if (p >= 0 && p < len) mask[static_cast<size_t>(p >> 5)] |= 1u << (p & 31);
    }
    if (int rc = d_mask.reserve_on_current_device(mask.size() * sizeof(uint32_t))) return rc;
    DV_HIP_CHECK(hipMemcpyAsync(d_mask.ptr, mask.data(), mask.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    DV_HIP_CHECK(hipStreamSynchronize(stream));      // `mask` is a local
    a.candidate_mask = static_cast<const uint32_t*>(d_mask.ptr);
    n_candidate_refs = static_cast<size_t>(o->n_candidate_positions);
  }
  if (int rc = d_cnt.reserve_on_current_device(static_cast<size_t>(len) * sizeof(int32_t))) return rc;
  if (int rc = d_ctr.reserve_on_current_device(4 * sizeof(uint32_t))) return rc;
  a.ref_count = static_cast<int32_t*>(d_cnt.ptr);
  a.counters = static_cast<uint32_t*>(d_ctr.ptr);
  // events: substitutions are a few per cent of the bases, indels at most one per CIGAR op;
  // the counter keeps counting past the capacity, so a second pass sizes it exactly
  uint32_t cap = b->n_cigar + b->n_bases / 16 + 4096 + static_cast<uint32_t>(std::min<size_t>(n_candidate_refs * 64, 1u << 24));
  uint32_t ctr[4] = {0, 0, 0, 0};
  for (int pass = 0; pass < 2; ++pass) {
    if (int rc = d_ev.reserve_on_current_device(static_cast<size_t>(cap) * sizeof(dv_allele_event))) return rc;
    a.events = static_cast<dv_allele_event*>(d_ev.ptr);
    a.event_cap = cap;
    DV_HIP_CHECK(hipMemsetAsync(d_cnt.ptr, 0, static_cast<size_t>(len) * sizeof(int32_t), stream));
    DV_HIP_CHECK(hipMemsetAsync(d_ctr.ptr, 0, 4 * sizeof(uint32_t), stream));
    {
      dv::ProfileScope prof(dv::kProfOther, stream);
      hipLaunchKernelGGL(count_alleles_kernel, dim3((b->n_reads + 4 * kReadsPerWave - 1) / (4 * kReadsPerWave)),
                         dim3(256), 0, stream, a);
    }
    DV_HIP_CHECK(hipGetLastError());
    DV_HIP_CHECK(hipMemcpyAsync(ctr, d_ctr.ptr, sizeof(ctr), hipMemcpyDeviceToHost, stream));
    DV_HIP_CHECK(hipStreamSynchronize(stream));
    if (ctr[0] <= cap) break;
    cap = ctr[0];
  }
  if (ctr[2] != 0) {
    return dv::fail(DV_ERR_BAD_INPUT,
                    "dv_count_alleles: an indel reaches outside the reference window (pass more margin)");
  }
  res->n_reads_counted = static_cast<int32_t>(ctr[1]);
  if (int rc = res->ref_count.reserve(static_cast<size_t>(len))) return rc;
  if (int rc = res->raw.reserve(ctr[0])) return rc;
  DV_HIP_CHECK(hipMemcpyAsync(res->ref_count.ptr, d_cnt.ptr, static_cast<size_t>(len) * sizeof(int32_t),
                              hipMemcpyDeviceToHost, stream));
  if (ctr[0]) {
    DV_HIP_CHECK(hipMemcpyAsync(res->raw.ptr, d_ev.ptr, static_cast<size_t>(ctr[0]) * sizeof(dv_allele_event),
                                hipMemcpyDeviceToHost, stream));
  }
  DV_HIP_CHECK(hipStreamSynchronize(stream));
  order_events(res.get(), ctr[0]);
  *out = res.release();
  return DV_OK;
}

// Several regions in one go (a region driver's batch of 1 kb calling regions: window selection of
// all of them, then their candidate counts).  What a one-region call spends is not the kernel (a few
// microseconds) but ~10 small pageable uploads, two stream synchronisations and two small
// downloads; here every region's arrays travel in ONE pinned staging image, the kernels are queued
// back to back, and the stream is synchronised twice for the whole batch.  Results per region are
// those of dv_count_alleles (the same kernel on the same arguments).  Regions whose tables are
// device-resident, and the rare region whose events overflow the first guess, take the one-region path.
int dv_count_alleles_batch(int32_t n, const dv_batch* const* reads, const dv_allele_counter_options* const* options,
                           dv_allele_counts** out, void* stream_v) {
  if (n < 0 || (n > 0 && (!reads || !options || !out))) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_count_alleles_batch: null");
  for (int32_t k = 0; k < n; ++k) out[k] = nullptr;
  auto fail_all = [&](int rc) {
    for (int32_t k = 0; k < n; ++k) {
      delete out[k];
      out[k] = nullptr;
    }
    return rc;
  };
  for (int32_t k = 0; k < n; ++k) {
    if (int rc = check_request(reads[k], options[k], &out[k], "dv_count_alleles_batch")) return rc;
    if (options[k]->track_ref_reads && options[k]->n_candidate_positions > 0 && !options[k]->candidate_positions) {
      return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_count_alleles_batch: candidate_positions is null");
    }
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  // Everything from here on may have results in out[]: the body runs as one callable so that EVERY error exit
  // -- the DV_HIP_CHECK returns included -- passes through fail_all ("on an error no result is left allocated").
  auto body = [&]() -> int {
  struct Plan {
    bool batched = false;
    size_t up[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // staging offsets: read_pos, seq_off, cigar_off, mapq, bases, quals, cigar, ref, mask
    size_t cnt_off = 0, ev_off = 0;              // in ints / events
    int64_t len = 0;
    uint32_t cap = 0;
    size_t mask_words = 0;
  };
  std::vector<Plan> plan(static_cast<size_t>(n));
  auto align16 = [](size_t x) { return (x + 15) & ~static_cast<size_t>(15); };
  size_t up_bytes = 0, cnt_ints = 0, ev_total = 0;
  int n_batched = 0;
  for (int32_t k = 0; k < n; ++k) {
    const dv_batch* b = reads[k];
    const dv_allele_counter_options* o = options[k];
    Plan& p = plan[k];
    p.len = o->interval_end - o->interval_start;
    if (b->n_reads == 0 || p.len == 0 || b->memory != DV_MEM_HOST) continue;     // one-region path below
    p.batched = true;
    ++n_batched;
    const size_t nr = static_cast<size_t>(b->n_reads);
    const size_t sizes[9] = {nr * 4, (nr + 1) * 4, (nr + 1) * 4, nr, b->n_bases, b->n_bases,
                             static_cast<size_t>(b->n_cigar) * 4, static_cast<size_t>(o->n_ref_bases), 0};
    for (int f = 0; f < 8; ++f) {
      p.up[f] = up_bytes;
      up_bytes += align16(std::max<size_t>(sizes[f], 1));
    }
    size_t n_candidate_refs = 0;
    if (o->track_ref_reads && o->n_candidate_positions > 0) {
      p.mask_words = static_cast<size_t>((p.len + 31) / 32);
      p.up[8] = up_bytes;
      up_bytes += align16(p.mask_words * 4);
      n_candidate_refs = static_cast<size_t>(o->n_candidate_positions);
    }
    p.cnt_off = cnt_ints;
    cnt_ints += static_cast<size_t>(p.len);
    p.cap = b->n_cigar + b->n_bases / 16 + 4096 + static_cast<uint32_t>(std::min<size_t>(n_candidate_refs * 64, 1u << 24));
    p.ev_off = ev_total;
    ev_total += p.cap;
  }
  if (n_batched > 0) {
    // grow-only scratch per host thread: pinned staging both ways, device images
    static thread_local PinnedBytes h_up, h_down;
    static thread_local dv::DeviceBuffer d_up, d_res, d_ev;
    const size_t ctr_bytes = align16(static_cast<size_t>(n) * 4 * sizeof(uint32_t));
    const size_t res_bytes = ctr_bytes + cnt_ints * sizeof(int32_t);
    if (int rc = h_up.reserve(up_bytes)) return rc;
    if (int rc = d_up.reserve_on_current_device(up_bytes)) return rc;
    if (int rc = d_res.reserve_on_current_device(res_bytes)) return rc;
    if (int rc = d_ev.reserve_on_current_device(std::max<size_t>(ev_total, 1) * sizeof(dv_allele_event))) return rc;
    for (int32_t k = 0; k < n; ++k) {
      const Plan& p = plan[k];
      if (!p.batched) continue;
      const dv_batch* b = reads[k];
      const dv_allele_counter_options* o = options[k];
      const size_t nr = static_cast<size_t>(b->n_reads);
      uint8_t* base = h_up.ptr;
      std::memcpy(base + p.up[0], b->read_pos, nr * 4);
      std::memcpy(base + p.up[1], b->read_seq_off, (nr + 1) * 4);
      std::memcpy(base + p.up[2], b->read_cigar_off, (nr + 1) * 4);
      std::memcpy(base + p.up[3], b->read_mapq, nr);
      std::memcpy(base + p.up[4], b->bases, b->n_bases);
      std::memcpy(base + p.up[5], b->quals, b->n_bases);
      std::memcpy(base + p.up[6], b->cigar, static_cast<size_t>(b->n_cigar) * 4);
      std::memcpy(base + p.up[7], o->ref_bases, static_cast<size_t>(o->n_ref_bases));
      if (p.mask_words) {
        uint32_t* mask = reinterpret_cast<uint32_t*>(base + p.up[8]);
        std::memset(mask, 0, p.mask_words * 4);
        for (int32_t c = 0; c < o->n_candidate_positions; ++c) {
          const int64_t q = o->candidate_positions[c] - o->interval_start;
          if (q >= 0 && q < p.len) mask[static_cast<size_t>(q >> 5)] |= 1u << (q & 31);
        }
      }
    }
    DV_HIP_CHECK(hipMemcpyAsync(d_up.ptr, h_up.ptr, up_bytes, hipMemcpyHostToDevice, stream));
    DV_HIP_CHECK(hipMemsetAsync(d_res.ptr, 0, res_bytes, stream));
    uint8_t* dres = static_cast<uint8_t*>(d_res.ptr);
    const uint8_t* dup = static_cast<const uint8_t*>(d_up.ptr);
    for (int32_t k = 0; k < n; ++k) {
      const Plan& p = plan[k];
      if (!p.batched) continue;
      const dv_batch* b = reads[k];
      const dv_allele_counter_options* o = options[k];
      CountArgs a{};
      a.n_reads = b->n_reads;
      a.read_pos = reinterpret_cast<const int32_t*>(dup + p.up[0]);
      a.seq_off = reinterpret_cast<const uint32_t*>(dup + p.up[1]);
      a.cigar_off = reinterpret_cast<const uint32_t*>(dup + p.up[2]);
      a.mapq = dup + p.up[3];
      a.bases = dup + p.up[4];
      a.quals = dup + p.up[5];
      a.cigar = reinterpret_cast<const uint32_t*>(dup + p.up[6]);
      a.ref = dup + p.up[7];
      a.ref_start = o->ref_start;
      a.n_ref = o->n_ref_bases;
      a.reads_start = std::min(o->interval_start, o->reads_interval_start);
      a.reads_end = std::max(o->interval_end, o->reads_interval_end);
      a.interval_start = o->interval_start;
      a.interval_len = p.len;
      a.contig_len = o->contig_n_bases > 0 ? o->contig_n_bases : o->ref_start + o->n_ref_bases;
      a.min_mapq = o->min_mapping_quality;
      a.min_bq = o->min_base_quality;
      a.legacy = o->keep_legacy_behavior ? 1 : 0;
      a.candidate_mask = p.mask_words ? reinterpret_cast<const uint32_t*>(dup + p.up[8]) : nullptr;
      a.counters = reinterpret_cast<uint32_t*>(dres) + static_cast<size_t>(k) * 4;
      a.ref_count = reinterpret_cast<int32_t*>(dres + ctr_bytes) + p.cnt_off;
      a.events = static_cast<dv_allele_event*>(d_ev.ptr) + p.ev_off;
      a.event_cap = p.cap;
      dv::ProfileScope prof(dv::kProfOther, stream);
      hipLaunchKernelGGL(count_alleles_kernel, dim3((b->n_reads + 4 * kReadsPerWave - 1) / (4 * kReadsPerWave)),
                         dim3(256), 0, stream, a);
    }
    DV_HIP_CHECK(hipGetLastError());
    // counters and counts come back together; the events follow once their numbers are known
    if (int rc = h_down.reserve(res_bytes)) return rc;
    DV_HIP_CHECK(hipMemcpyAsync(h_down.ptr, d_res.ptr, res_bytes, hipMemcpyDeviceToHost, stream));
    DV_HIP_CHECK(hipStreamSynchronize(stream));
    const uint32_t* ctrs = reinterpret_cast<const uint32_t*>(h_down.ptr);
    const int32_t* cnts = reinterpret_cast<const int32_t*>(h_down.ptr + ctr_bytes);
    size_t ev_bytes = 0;
    for (int32_t k = 0; k < n; ++k) {
      Plan& p = plan[k];
      if (!p.batched) continue;
      const uint32_t* ctr = ctrs + static_cast<size_t>(k) * 4;
      if (ctr[2] != 0) {
        return fail_all(dv::fail(DV_ERR_BAD_INPUT,
                                 "dv_count_alleles: an indel reaches outside the reference window (pass more margin)"));
      }
      if (ctr[0] > p.cap) {          // the first guess was too small: this region goes alone (two passes there)
        p.batched = false;
        continue;
      }
      auto res = std::make_unique<dv_allele_counts>();
      res->length = p.len;
      res->n_reads_counted = static_cast<int32_t>(ctr[1]);
      if (int rc = res->ref_count.reserve(static_cast<size_t>(p.len))) return fail_all(rc);
      std::memcpy(res->ref_count.ptr, cnts + p.cnt_off, static_cast<size_t>(p.len) * sizeof(int32_t));
      if (int rc = res->raw.reserve(ctr[0])) return fail_all(rc);
      out[k] = res.release();
      ev_bytes += static_cast<size_t>(ctr[0]) * sizeof(dv_allele_event);
    }
    // the events of every region into one pinned image, one synchronisation
    static thread_local PinnedBytes h_ev;
    if (int rc = h_ev.reserve(ev_bytes)) return fail_all(rc);
    size_t at = 0;
    for (int32_t k = 0; k < n; ++k) {
      const Plan& p = plan[k];
      if (!p.batched || !out[k]) continue;
      const size_t bytes = static_cast<size_t>(ctrs[static_cast<size_t>(k) * 4]) * sizeof(dv_allele_event);
      if (bytes) {
        DV_HIP_CHECK(hipMemcpyAsync(h_ev.ptr + at, static_cast<const dv_allele_event*>(d_ev.ptr) + p.ev_off, bytes,
                                    hipMemcpyDeviceToHost, stream));
      }
      at += bytes;
    }
    DV_HIP_CHECK(hipStreamSynchronize(stream));
    at = 0;
    for (int32_t k = 0; k < n; ++k) {
      const Plan& p = plan[k];
      if (!p.batched || !out[k]) continue;
      const size_t n_ev = ctrs[static_cast<size_t>(k) * 4];
      if (n_ev) std::memcpy(out[k]->raw.ptr, h_ev.ptr + at, n_ev * sizeof(dv_allele_event));
      at += n_ev * sizeof(dv_allele_event);
      order_events(out[k], n_ev);
    }
  }
  for (int32_t k = 0; k < n; ++k) {
    if (out[k]) continue;
    if (int rc = dv_count_alleles(reads[k], options[k], &out[k], stream_v)) return rc;
  }
  return DV_OK;
  };
  const int rc = body();
  return rc == DV_OK ? DV_OK : fail_all(rc);
}

int dv_allele_counts_arrays(const dv_allele_counts* c, const int32_t** ref_supporting_read_count,
                            const dv_allele_event** events, uint32_t* n_events, int32_t* n_reads_counted) {
  if (!c) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_allele_counts_arrays: null");
  if (ref_supporting_read_count) *ref_supporting_read_count = c->ref_count.ptr ? c->ref_count.ptr : c->empty_counts.data();
  if (events) *events = c->events.data();
  if (n_events) *n_events = static_cast<uint32_t>(c->events.size());
  if (n_reads_counted) *n_reads_counted = c->n_reads_counted;
  return static_cast<int>(c->length);
}

void dv_allele_counts_free(dv_allele_counts* c) { delete c; }

}  // extern "C"
