// common.hip -- error handling, device buffers, profiling hooks and the small
// host-side helpers of the C ABI (CRC32C, DownsampleReadIndices, Query).
#include <algorithm>
#include <atomic>
#include <mutex>
#include <cstring>
#include <numeric>
#include <random>

#include "dv_internal.h"

namespace dv {

static thread_local std::string g_error;

void set_error(const std::string& msg) { g_error = msg; }

int fail(int status, const std::string& msg) {
  g_error = msg;
  return status;
}

// For function-local static / thread_local scratch, which outlives a hipSetDevice: a buffer belongs to the
// device that was current when it was allocated, and memory of another device must not be handed to a kernel
// of this one.  (Object-owned buffers -- a model's, an encoder's -- call reserve(): their owner sets the device.)
int DeviceBuffer::reserve_on_current_device(size_t bytes) {
  int now = -1;
  if (ptr != nullptr && hipGetDevice(&now) == hipSuccess && now != device) release();
  return reserve(bytes);
}

int DeviceBuffer::reserve(size_t bytes) {
  if (bytes <= cap) return DV_OK;
  release();
  (void)hipGetDevice(&device);
  size_t want = std::max<size_t>(bytes + bytes / 4, 256);
  hipError_t e = hipMalloc(&ptr, want);
  if (e != hipSuccess) {
    ptr = nullptr;
    return fail(DV_ERR_OUT_OF_MEMORY,
                std::string("hipMalloc: ") + hipGetErrorString(e));
  }
  cap = want;
  return DV_OK;
}

void DeviceBuffer::release() {
  if (ptr) (void)hipFree(ptr);
  ptr = nullptr;
  cap = 0;
}

// ---- profiling: event pairs around launches, summed per kind -------------
// Scopes on different host threads record concurrently: a scope holds its own pair and the
// shared pool / per-kind lists are only touched under g_prof_mu.
static std::atomic<bool> g_profiling{false};
static std::mutex g_prof_mu;
static std::vector<ProfileEvents> g_events[kProfKinds];
static std::vector<ProfileEvents> g_pool;

bool profiling_enabled() { return g_profiling.load(std::memory_order_relaxed); }

ProfileEvents profile_begin(hipStream_t stream) {
  ProfileEvents p;
  {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (!g_pool.empty()) {
      p = g_pool.back();
      g_pool.pop_back();
    }
  }
  if (!p.a) {
    (void)hipEventCreate(&p.a);
    (void)hipEventCreate(&p.b);
  }
  (void)hipEventRecord(p.a, stream);
  return p;
}

void profile_end(int kind, const ProfileEvents& ev, hipStream_t stream) {
  (void)hipEventRecord(ev.b, stream);
  std::lock_guard<std::mutex> lock(g_prof_mu);
  g_events[kind].push_back(ev);
}

}  // namespace dv

extern "C" {

const char* dv_last_error(void) { return dv::g_error.c_str(); }

int dv_abi_version(void) { return DV_ABI_VERSION; }

int dv_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int dv_set_profiling(int enabled) {
  dv::g_profiling.store(enabled != 0);
  std::lock_guard<std::mutex> lock(dv::g_prof_mu);
  for (int k = 0; k < dv::kProfKinds; ++k) {
    for (auto& p : dv::g_events[k]) dv::g_pool.push_back(p);
    dv::g_events[k].clear();
  }
  return DV_OK;
}

// Sums and clears the recorded launches of `kind`; also returns the launch
// count through dv_last_profile_count() (of the calling thread's last dv_profile_ms).
static thread_local int g_last_count = 0;

double dv_profile_ms(int kind) {
  if (kind < 0 || kind >= dv::kProfKinds) return 0.0;
  std::vector<dv::ProfileEvents> mine;
  {
    std::lock_guard<std::mutex> lock(dv::g_prof_mu);
    mine.swap(dv::g_events[kind]);
  }
  double total = 0.0;
  g_last_count = 0;
  for (auto& p : mine) {
    if (hipEventSynchronize(p.b) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      total += ms;
      ++g_last_count;
    }
  }
  std::lock_guard<std::mutex> lock(dv::g_prof_mu);
  for (auto& p : mine) dv::g_pool.push_back(p);
  return total;
}

int dv_last_profile_count(void) { return g_last_count; }

// CRC32C, slicing-by-8 tables (Castagnoli polynomial, reflected 0x82F63B78).
uint32_t dv_crc32c(const uint8_t* data, size_t n) {
  static uint32_t table[8][256];
  static std::once_flag once;
  std::call_once(once, [] {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      table[0][i] = c;
    }
    for (int k = 1; k < 8; ++k)
      for (uint32_t i = 0; i < 256; ++i)
        table[k][i] = (table[k - 1][i] >> 8) ^ table[0][table[k - 1][i] & 0xFF];
  });
  uint32_t crc = 0xFFFFFFFFu;
  while (n >= 8) {
    uint32_t lo, hi;
    memcpy(&lo, data, 4);
    memcpy(&hi, data + 4, 4);
    lo ^= crc;
    crc = table[7][lo & 0xFF] ^ table[6][(lo >> 8) & 0xFF] ^
          table[5][(lo >> 16) & 0xFF] ^ table[4][lo >> 24] ^
          table[3][hi & 0xFF] ^ table[2][(hi >> 8) & 0xFF] ^
          table[1][(hi >> 16) & 0xFF] ^ table[0][hi >> 24];
    data += 8;
    n -= 8;
  }
  while (n--) crc = (crc >> 8) ^ table[0][(crc ^ *data++) & 0xFF];
  return crc ^ 0xFFFFFFFFu;
}

// pileup_image_native.cc:153-165.  std::shuffle from the same libstdc++ the
// reference is built against; the generator restarts per call.
int dv_downsample_indices(int n, int max_reads, uint32_t seed, int32_t* out) {
  if (n < 0 || out == nullptr) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_downsample_indices: bad args");
  }
  std::vector<int> idx(n);
  std::iota(idx.begin(), idx.end(), 0);
  if (n > max_reads) {
    std::mt19937_64 gen(seed);
    std::shuffle(idx.begin(), idx.end(), gen);
  }
  for (int i = 0; i < n; ++i) out[i] = idx[i];
  return DV_OK;
}

// make_examples_native.cc:802-810 + nucleus/util/utils.cc:172-240.
int dv_query_reads(int32_t n_reads, const int32_t* read_pos,
                   const uint32_t* read_cigar_off, const uint32_t* cigar,
                   int32_t n_items, const int64_t* query_start,
                   const int64_t* query_end, uint32_t* list_off,
                   uint32_t* list_read) {
  if (n_reads < 0 || n_items < 0 || !list_off) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_query_reads: bad args");
  }
  std::vector<int64_t> read_end(n_reads);
  for (int r = 0; r < n_reads; ++r) {
    int64_t e = read_pos[r];
    for (uint32_t c = read_cigar_off[r]; c < read_cigar_off[r + 1]; ++c) {
      const uint32_t op = cigar[c] & 0xF;
      if (op == DV_CIGAR_ALIGNMENT_MATCH || op == DV_CIGAR_SEQUENCE_MATCH ||
          op == DV_CIGAR_DELETE || op == DV_CIGAR_SKIP ||
          op == DV_CIGAR_SEQUENCE_MISMATCH) {
        e += cigar[c] >> 4;
      }
    }
    read_end[r] = e;
  }
  uint32_t total = 0;
  for (int i = 0; i < n_items; ++i) {
    list_off[i] = total;
    for (int r = 0; r < n_reads; ++r) {
      if (query_end[i] > read_pos[r] && query_start[i] < read_end[r]) {
        if (list_read) list_read[total] = r;
        ++total;
      }
    }
  }
  list_off[n_items] = total;
  return DV_OK;
}

}  // extern "C"
