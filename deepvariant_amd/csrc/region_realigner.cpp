// dv_realign_regions (include/dvhip.h, "the window realigner over many regions"): the body of
// Realigner.realign_reads (deepvariant/realigner/realigner.py:795-855) for a batch of calling
// regions, with the (region, window) tasks spread over host threads.
//
//   phase 1, one task per candidate window: the reads that overlap the window, in row order,
//            go into a de Bruijn graph with the window's reference (call_debruijn_graph,
//            :703-738); a window survives when its haplotypes are not just the reference;
//   between: per region every read joins the surviving window it shares most bases with,
//            the first one on ties (assign_reads_to_assembled_regions, :596-619);
//   phase 2, one task per surviving window with reads: FastPassAligner::AlignReads against
//            the haplotypes padded with the reference out to the reads' span + margin
//            (call_fast_pass_aligner, :740-793).
//
// Tasks are independent (each builds its own graph / aligner; the aligner's scratch buffers are
// thread_local), so a phase is a parallel loop over a task list ordered longest first.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

#include "debruijn_graph.h"
#include "dv_internal.h"
#include "fast_pass_aligner.h"

struct dv_realign_result {
  std::vector<int64_t> region_row_off;
  std::vector<int32_t> order, status;
  std::vector<int64_t> position, cigar_off;
  std::vector<uint32_t> cigar;
  std::vector<int32_t> region_assembled_off, assembled_window, assembled_hap_off;
  std::vector<int64_t> hap_text_off;
  std::string hap_text;
};

namespace {

struct Window {            // one candidate window of one region
  int region = 0, index = 0;
  std::vector<std::string> haplotypes;   // empty: dropped
  std::vector<int32_t> rows;             // the reads it claimed, ascending
  std::vector<dv::RealignedRead> aligned;   // parallel to rows; empty: alignments kept
};

// The worker threads live as long as the library: no thread creation per call, and the aligner's
// thread_local scratch tables (MBs for a heavy window) stay allocated from one call to the next.
// One job at a time; the calling thread works too.
class WorkerPool {
 public:
  static WorkerPool& get() {
    static WorkerPool* pool = new WorkerPool();      // never destroyed: workers may outlive main()'s statics
    return *pool;
  }

  void run(const std::vector<int>& tasks, int n_threads, const std::function<void(int)>& fn) {
    const int n = static_cast<int>(tasks.size());
    if (n_threads <= 1 || n <= 1) {
      for (int t : tasks) fn(t);
      return;
    }
    std::lock_guard<std::mutex> one_job(job_lock_);
    const int helpers = std::min(n_threads, n) - 1;
    {
      std::lock_guard<std::mutex> hold(lock_);
      if (pid_ != getpid()) {
        // after fork() the helper threads do not exist in the child: forget them (their std::thread objects are
        // leaked on purpose -- destroying a joinable one terminates) and start over with this process's own
        pid_ = getpid();
        (void)new std::vector<std::thread>(std::move(threads_));
        threads_.clear();
        wanted_ = busy_ = 0;
      }
      error_.clear();
      while (static_cast<int>(threads_.size()) < helpers) threads_.emplace_back([this] { worker(); });
      tasks_ = &tasks;
      fn_ = &fn;
      next_.store(0, std::memory_order_relaxed);
      wanted_ = helpers;
      busy_ = helpers;
      ++generation_;
    }
    wake_.notify_all();
    drain();
    std::unique_lock<std::mutex> hold(lock_);
    done_.wait(hold, [this] { return busy_ == 0; });
    tasks_ = nullptr;
    fn_ = nullptr;
    if (!error_.empty()) throw std::runtime_error(error_);   // on the CALLING thread, after every helper has let go
  }

 private:
  void drain() {
    const int n = static_cast<int>(tasks_->size());
    for (;;) {
      const int i = next_.fetch_add(1, std::memory_order_relaxed);
      if (i >= n) return;
      // an exception (std::bad_alloc, length_error ...) must neither leave a helper thread (std::terminate) nor
      // unwind the caller while helpers still hold tasks_ / fn_: the first one is kept, the rest of the job is
      // skipped, run() rethrows it once everybody is done
      try {
        (*fn_)((*tasks_)[i]);
      } catch (const std::exception& e) {
        fail_job(e.what());
      } catch (...) {
        fail_job("unknown exception in a realigner task");
      }
    }
  }

  void fail_job(const char* what) {
    std::lock_guard<std::mutex> hold(lock_);
    if (error_.empty()) error_ = what;
    next_.store(1 << 30, std::memory_order_relaxed);
  }

  void worker() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> hold(lock_);
        wake_.wait(hold, [&] { return generation_ != seen && wanted_ > 0; });
        seen = generation_;
        --wanted_;
      }
      drain();
      {
        std::lock_guard<std::mutex> hold(lock_);
        if (--busy_ == 0) done_.notify_one();
      }
    }
  }

  std::mutex job_lock_, lock_;
  std::condition_variable wake_, done_;
  std::vector<std::thread> threads_;
  const std::vector<int>* tasks_ = nullptr;
  const std::function<void(int)>* fn_ = nullptr;
  std::atomic<int> next_{0};
  int wanted_ = 0, busy_ = 0;
  uint64_t generation_ = 0;
  std::string error_;
  pid_t pid_ = getpid();
};

void parallel_tasks(const std::vector<int>& tasks, int n_threads, const std::function<void(int)>& fn) {
  WorkerPool::get().run(tasks, n_threads, fn);
}

std::vector<int> longest_first(const std::vector<int64_t>& cost) {
  std::vector<int> order(cost.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = static_cast<int>(i);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
  return order;
}

}  // namespace

extern "C" {

static int realign_regions_impl(const dv_realign_region* regions, int32_t n_regions, const dv_realign_options* o,
                                dv_realign_result** out, dv_realign_output* arrays);

// The exception barrier of the entry point: nothing unwinds through extern "C" into ctypes.
int dv_realign_regions(const dv_realign_region* regions, int32_t n_regions, const dv_realign_options* o,
                       dv_realign_result** out, dv_realign_output* arrays) {
  try {
    return realign_regions_impl(regions, n_regions, o, out, arrays);
  } catch (const std::bad_alloc&) {
    if (out) *out = nullptr;
    return dv::fail(DV_ERR_OUT_OF_MEMORY, "dv_realign_regions: out of host memory");
  } catch (const std::exception& e) {
    if (out) *out = nullptr;
    return dv::fail(DV_ERR_BAD_INPUT, std::string("dv_realign_regions: ") + e.what());
  }
}

static int realign_regions_impl(const dv_realign_region* regions, int32_t n_regions, const dv_realign_options* o,
                                dv_realign_result** out, dv_realign_output* arrays) {
  if (!out || !arrays || !o || n_regions < 0 || (n_regions > 0 && !regions)) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_realign_regions: null argument");
  }
  *out = nullptr;
  if (o->dbg.step_k <= 0 || o->dbg.min_k <= 0) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_realign_regions: min_k and step_k must be positive");
  }
  if (o->ref_align_margin < 0) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_realign_regions: negative margin");
  std::vector<Window> windows;
  std::vector<int64_t> cost;
  for (int32_t g = 0; g < n_regions; ++g) {
    const dv_realign_region& r = regions[g];
    if (r.n_reads < 0 || r.n_windows < 0 || r.ref_len < 0 || r.ref_start < 0 || r.n_bases < 0 ||
        (r.n_reads > 0 && (!r.bases || !r.quals || !r.read_seq_off || !r.read_mapq || !r.read_start || !r.read_end)) ||
        (r.n_windows > 0 && (!r.window_start || !r.window_end || !r.ref))) {
      return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_realign_regions: null or negative field in a region");
    }
    for (int32_t i = 0; i < r.n_reads; ++i) {
      if (r.read_seq_off[i + 1] < r.read_seq_off[i] || r.read_seq_off[i + 1] > r.n_bases) {
        return dv::fail(DV_ERR_BAD_INPUT, "dv_realign_regions: a read's bases lie outside the table");
      }
    }
    for (int32_t w = 0; w < r.n_windows; ++w) {
      const int64_t a = r.window_start[w], b = r.window_end[w];
      if (a < 0 || a > b || a >= r.contig_len || b > r.contig_len) {
        return dv::fail(DV_ERR_BAD_INPUT, "dv_realign_regions: window outside the contig");
      }
      if (a < r.ref_start || b > r.ref_start + r.ref_len) {
        return dv::fail(DV_ERR_BAD_INPUT, "dv_realign_regions: window outside the reference bases handed over");
      }
      Window win;
      win.region = g;
      win.index = w;
      windows.push_back(std::move(win));
      cost.push_back((b - a) * static_cast<int64_t>(r.n_reads + 1));
    }
  }
  int n_threads = o->n_threads;
  if (n_threads <= 0) n_threads = static_cast<int>(std::min(16u, std::max(1u, std::thread::hardware_concurrency())));

  dv::DeBruijnOptions dbg;
  dbg.min_k = o->dbg.min_k;
  dbg.max_k = o->dbg.max_k;
  dbg.step_k = o->dbg.step_k;
  dbg.min_mapq = o->dbg.min_mapq;
  dbg.min_base_quality = o->dbg.min_base_quality;
  dbg.min_edge_weight = o->dbg.min_edge_weight;
  dbg.max_num_paths = o->dbg.max_num_paths;
  dbg.disable_graph_pruning = o->dbg.disable_graph_pruning != 0;

  // ---- phase 1: assembly
  parallel_tasks(longest_first(cost), n_threads, [&](int t) {
    Window& win = windows[t];
    const dv_realign_region& r = regions[win.region];
    const int64_t a = r.window_start[win.index], b = r.window_end[win.index];
    const std::string_view ref(r.ref + (a - r.ref_start), static_cast<size_t>(b - a));
    std::vector<dv::AssemblyRead> reads;
    for (int32_t i = 0; i < r.n_reads; ++i) {
      if (r.read_end[i] > a && b > r.read_start[i]) {
        const uint32_t s0 = r.read_seq_off[i], s1 = r.read_seq_off[i + 1];
        reads.push_back(dv::AssemblyRead{std::string_view(reinterpret_cast<const char*>(r.bases) + s0, s1 - s0),
                                         r.quals + s0, r.read_mapq[i]});
      }
    }
    auto graph = dv::DeBruijnGraph::build(ref, reads, dbg);
    if (!graph) return;                                   // haplotypes == [ref]: dropped
    std::vector<std::string> haps = graph->candidate_haplotypes();
    if (haps.empty() || (haps.size() == 1 && haps[0] == ref)) return;
    win.haplotypes = std::move(haps);
  });

  // ---- reads -> assembled windows (per region; windows of a region are consecutive in `windows`)
  auto res = std::make_unique<dv_realign_result>();
  res->region_row_off.assign(1, 0);
  for (int32_t g = 0; g < n_regions; ++g) res->region_row_off.push_back(res->region_row_off.back() + regions[g].n_reads);
  const int64_t total_rows = res->region_row_off.back();
  std::vector<int32_t> row_window(static_cast<size_t>(total_rows), -1);   // index into `windows`
  {
    size_t w0 = 0;
    for (int32_t g = 0; g < n_regions; ++g) {
      const dv_realign_region& r = regions[g];
      const size_t w1 = w0 + static_cast<size_t>(r.n_windows);
      std::vector<size_t> alive;
      for (size_t w = w0; w < w1; ++w) {
        if (!windows[w].haplotypes.empty()) alive.push_back(w);
      }
      if (!alive.empty()) {
        const int64_t base = res->region_row_off[g];
        for (int32_t i = 0; i < r.n_reads; ++i) {
          int64_t best = 0;
          size_t best_w = 0;
          for (size_t w : alive) {
            const int64_t shared = std::min(r.read_end[i], r.window_end[windows[w].index]) -
                                   std::max(r.read_start[i], r.window_start[windows[w].index]);
            if (shared > best) {       // strictly more: the first window wins ties
              best = shared;
              best_w = w;
            }
          }
          if (best > 0) {
            row_window[base + i] = static_cast<int32_t>(best_w);
            windows[best_w].rows.push_back(i);
          }
        }
      }
      w0 = w1;
    }
  }

  // ---- phase 2: alignment
  std::vector<int> align_tasks;
  std::vector<int64_t> align_cost;
  for (size_t w = 0; w < windows.size(); ++w) {
    if (windows[w].rows.empty()) continue;
    align_tasks.push_back(static_cast<int>(w));
    int64_t hap_bases = 0;
    for (const std::string& h : windows[w].haplotypes) hap_bases += static_cast<int64_t>(h.size()) + 300;
    align_cost.push_back(hap_bases * static_cast<int64_t>(windows[w].haplotypes.size() + windows[w].rows.size()));
  }
  {
    std::vector<int> by_cost = longest_first(align_cost);
    for (int& t : by_cost) t = align_tasks[t];
    align_tasks = std::move(by_cost);
  }
  std::mutex error_lock;
  std::string error;
  parallel_tasks(align_tasks, n_threads, [&](int t) {
    Window& win = windows[t];
    const dv_realign_region& r = regions[win.region];
    const int64_t a = r.window_start[win.index], b = r.window_end[win.index];
    int64_t lo = r.read_start[win.rows[0]], hi = r.read_end[win.rows[0]];
    for (int32_t i : win.rows) {
      lo = std::min(lo, r.read_start[i]);
      hi = std::max(hi, r.read_end[i]);
    }
    const int64_t ref_start = std::max<int64_t>(0, std::min(lo, a) - o->ref_align_margin);
    const int64_t ref_end = std::min(r.contig_len, std::max(hi, b) + o->ref_align_margin);
    if (ref_end <= b) return;            // no room for a suffix: the original alignments stay
    if (ref_start < r.ref_start || ref_end > r.ref_start + r.ref_len) {
      std::lock_guard<std::mutex> hold(error_lock);
      error = "dv_realign_regions: reads of a window reach outside the reference bases handed over";
      return;
    }
    const std::string prefix(r.ref + (ref_start - r.ref_start), static_cast<size_t>(a - ref_start));
    const std::string window_ref(r.ref + (a - r.ref_start), static_cast<size_t>(b - a));
    const std::string suffix(r.ref + (b - r.ref_start), static_cast<size_t>(ref_end - b));
    std::vector<std::string> haplotypes;
    haplotypes.reserve(win.haplotypes.size());
    for (const std::string& h : win.haplotypes) {
      haplotypes.push_back(prefix + h + suffix);
      if (haplotypes.back().size() >= 0xffff) {
        std::lock_guard<std::mutex> hold(error_lock);
        error = "haplotypes are limited to 65534 bases (16-bit read offsets)";
        return;
      }
    }
    std::vector<std::string> sequences;
    sequences.reserve(win.rows.size());
    for (int32_t i : win.rows) {
      sequences.emplace_back(reinterpret_cast<const char*>(r.bases) + r.read_seq_off[i],
                             r.read_seq_off[i + 1] - r.read_seq_off[i]);
    }
    dv::AlignerOptions ao;
    ao.match = o->aln.match;
    ao.mismatch = o->aln.mismatch;
    ao.gap_open = o->aln.gap_open;
    ao.gap_extend = o->aln.gap_extend;
    ao.kmer_size = o->aln.kmer_size;
    ao.read_size = static_cast<int>(sequences[0].size());
    ao.max_num_of_mismatches = o->aln.max_num_of_mismatches;
    ao.similarity_threshold = o->aln.realignment_similarity_threshold;
    ao.force_alignment = false;
    dv::FastPassAligner aligner;
    std::string why;
    if (!aligner.set_options(ao, &why)) {
      std::lock_guard<std::mutex> hold(error_lock);
      error = why;
      return;
    }
    aligner.set_normalize_reads(o->aln.normalize_reads != 0);
    aligner.set_ref_prefix_len(static_cast<int>(prefix.size()));
    aligner.set_ref_suffix_len(static_cast<int>(suffix.size()));
    aligner.set_reference(prefix + window_ref + suffix);
    aligner.set_ref_start(static_cast<uint64_t>(ref_start));
    aligner.set_haplotypes(haplotypes);
    win.aligned = aligner.align_reads(sequences);
  });
  if (!error.empty()) return dv::fail(DV_ERR_BAD_INPUT, error);

  // ---- results
  res->order.resize(static_cast<size_t>(total_rows));
  res->status.assign(static_cast<size_t>(total_rows), 0);
  res->position.assign(static_cast<size_t>(total_rows), 0);
  res->cigar_off.assign(static_cast<size_t>(total_rows) + 1, 0);
  res->region_assembled_off.assign(1, 0);
  res->assembled_hap_off.assign(1, 0);
  res->hap_text_off.assign(1, 0);
  std::vector<const dv::Cigar*> row_cigar(static_cast<size_t>(total_rows), nullptr);
  size_t w0 = 0;
  for (int32_t g = 0; g < n_regions; ++g) {
    const dv_realign_region& r = regions[g];
    const size_t w1 = w0 + static_cast<size_t>(r.n_windows);
    const int64_t base = res->region_row_off[g];
    int64_t at = base;
    for (int32_t i = 0; i < r.n_reads; ++i) {
      if (row_window[base + i] < 0) res->order[at++] = i;
    }
    for (size_t w = w0; w < w1; ++w) {
      const Window& win = windows[w];
      if (win.haplotypes.empty()) continue;
      res->assembled_window.push_back(win.index);
      for (const std::string& h : win.haplotypes) {
        res->hap_text += h;
        res->hap_text_off.push_back(static_cast<int64_t>(res->hap_text.size()));
      }
      res->assembled_hap_off.push_back(static_cast<int32_t>(res->hap_text_off.size() - 1));
      for (size_t k = 0; k < win.rows.size(); ++k) {
        const int32_t i = win.rows[k];
        res->order[at++] = i;
        if (!win.aligned.empty() && win.aligned[k].status == 1) {
          res->status[base + i] = 1;
          res->position[base + i] = win.aligned[k].position;
          row_cigar[base + i] = &win.aligned[k].cigar;
        }
      }
    }
    res->region_assembled_off.push_back(static_cast<int32_t>(res->assembled_window.size()));
    w0 = w1;
  }
  for (int64_t i = 0; i < total_rows; ++i) {
    if (row_cigar[i]) {
      for (const dv::CigarOp& op : *row_cigar[i]) {
        res->cigar.push_back((static_cast<uint32_t>(op.length) << 4) | static_cast<uint32_t>(op.op));
      }
    }
    res->cigar_off[i + 1] = static_cast<int64_t>(res->cigar.size());
  }
  arrays->region_row_off = res->region_row_off.data();
  arrays->order = res->order.data();
  arrays->status = res->status.data();
  arrays->position = res->position.data();
  arrays->cigar_off = res->cigar_off.data();
  arrays->cigar = res->cigar.data();
  arrays->region_assembled_off = res->region_assembled_off.data();
  arrays->assembled_window = res->assembled_window.data();
  arrays->assembled_hap_off = res->assembled_hap_off.data();
  arrays->hap_text_off = res->hap_text_off.data();
  arrays->hap_text = res->hap_text.data();
  *out = res.release();
  return DV_OK;
}

void dv_realign_result_free(dv_realign_result* r) { delete r; }

}  // extern "C"
