// See local_align.h.
#include "local_align.h"

#include <algorithm>
#include <cstdlib>

namespace dv {
namespace {

inline int8_t base_code(char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
  }
}

std::vector<int8_t> translate(const std::string& s) {
  std::vector<int8_t> out(s.size());
  for (size_t i = 0; i < s.size(); ++i) out[i] = base_code(s[i]);
  return out;
}

// ---- the sweep for 16 (reference, query) pairs at once --------------------------------------
// Lane l runs sweep()'s recurrence for pair l: same H, E, F, the same "first column whose
// maximum exceeds every earlier one" and the same saved column.  Values fit int16: H <= match *
// |query| (checked by the caller), E and F never drop below -gap_open because H >= 0.  Lanes
// walk their sequences with a step of +1 or -1 (the reverse pass), shorter ones are padded with
// symbols that match nothing: a padded cell is always strictly below a real cell that was seen
// earlier, so it can neither start a new maximum nor be picked as its row.
// 16 lanes fill an AVX2 register, 32 an AVX-512 one; which one runs is a property of the host
// (sweep_width), the result per pair is the same either way.
constexpr int kMaxLanes = 32;
template <int L> struct LaneVector { typedef int16_t type __attribute__((vector_size(2 * L))); };

struct SweepLane {
  const int8_t* ref = nullptr;
  int ref_step = 1, ref_len = 0;
  const int8_t* q = nullptr;
  int q_step = 1, q_len = 0;
  int stop_at = -1;                     // the reverse pass stops once this score is reached
  int best = 0, best_ref = -1, best_q = -1;   // indices in walking order
};

struct SweepParams {
  int match, mismatch, gap_open, gap_extend;
};

template <int kLanes>
static inline __attribute__((always_inline)) void sweep_lanes_body(SweepLane* lanes, const SweepParams& p) {
  typedef typename LaneVector<kLanes>::type Lanes;
  int n = 0, columns = 0;
  for (int l = 0; l < kLanes; ++l) {
    n = std::max(n, lanes[l].q_len);
    columns = std::max(columns, lanes[l].ref_len);
  }
  const Lanes zero = {};
  const Lanes match = zero + static_cast<int16_t>(p.match), miss = zero - static_cast<int16_t>(p.mismatch);
  const Lanes go = zero + static_cast<int16_t>(p.gap_open), ge = zero + static_cast<int16_t>(p.gap_extend);
  static thread_local std::vector<Lanes> qv, prev, cur, e_col, best_col;
  qv.resize(n);
  prev.assign(n, zero);
  cur.assign(n, zero);
  e_col.assign(n, zero);
  best_col.assign(n, zero);
  Lanes stop = zero - static_cast<int16_t>(1);
  for (int l = 0; l < kLanes; ++l) {
    const SweepLane& a = lanes[l];
    stop[l] = static_cast<int16_t>(a.ref_len > 0 ? a.stop_at : 0);    // an empty lane is "done" at score 0
    for (int j = 0; j < n; ++j) {
      const int8_t c = j < a.q_len ? a.q[j * a.q_step] : 4;
      qv[j][l] = c < 4 ? c : 5;                                          // 'N' and padding match nothing
    }
  }
  Lanes best = zero, best_ref = zero - static_cast<int16_t>(1);
  for (int i = 0; i < columns; ++i) {
    Lanes r;
    for (int l = 0; l < kLanes; ++l) r[l] = i < lanes[l].ref_len ? lanes[l].ref[i * lanes[l].ref_step] : 4;
    Lanes f = zero, diag = zero, col_max = zero;
    for (int j = 0; j < n; ++j) {
      const Lanes eq = r == qv[j];
      Lanes h = diag + ((eq & match) | (~eq & miss));
      const Lanes e = e_col[j];
      h = __builtin_elementwise_max(__builtin_elementwise_max(h, e), __builtin_elementwise_max(f, zero));
      diag = prev[j];
      cur[j] = h;
      col_max = __builtin_elementwise_max(col_max, h);
      const Lanes open = h - go;
      e_col[j] = __builtin_elementwise_max(e - ge, open);
      f = __builtin_elementwise_max(f - ge, open);
    }
    prev.swap(cur);
    const Lanes improved = col_max > best;
    bool any = false;
    for (int l = 0; l < kLanes; ++l) any = any || improved[l];
    if (any) {
      best = __builtin_elementwise_max(best, col_max);
      best_ref = (improved & (zero + static_cast<int16_t>(i))) | (~improved & best_ref);
      for (int j = 0; j < n; ++j) best_col[j] = (improved & prev[j]) | (~improved & best_col[j]);
      const Lanes done = best == stop;
      bool all = true;
      for (int l = 0; l < kLanes; ++l) all = all && done[l];
      if (all) break;
    }
  }
  for (int l = 0; l < kLanes; ++l) {
    SweepLane& a = lanes[l];
    a.best = best[l];
    a.best_ref = best_ref[l];
    a.best_q = -1;
    for (int j = 0; j < a.q_len && a.best > 0; ++j) {
      if (best_col[j][l] == best[l]) {
        a.best_q = j;
        break;
      }
    }
  }
}

__attribute__((target("avx512bw,avx512vl,avx512f"))) void sweep_lanes_avx512(SweepLane* lanes, const SweepParams& p) {
  sweep_lanes_body<32>(lanes, p);
}
__attribute__((target("avx2"))) void sweep_lanes_avx2(SweepLane* lanes, const SweepParams& p) {
  sweep_lanes_body<16>(lanes, p);
}
void sweep_lanes_generic(SweepLane* lanes, const SweepParams& p) { sweep_lanes_body<16>(lanes, p); }

// pairs per sweep on this host: 32 with AVX-512BW, else 16 (DV_SWEEP_LANES=16 forces the narrow form)
int sweep_width() {
  static const int width = [] {
    const char* forced = getenv("DV_SWEEP_LANES");
    if (forced && std::atoi(forced) == 16) return 16;
    return __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") ? 32 : 16;
  }();
  return width;
}

void sweep_lanes(SweepLane* lanes, const SweepParams& p, int width) {
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (width == 32) {
    sweep_lanes_avx512(lanes, p);
  } else if (avx2) {
    sweep_lanes_avx2(lanes, p);
  } else {
    sweep_lanes_generic(lanes, p);
  }
}

}  // namespace

CodedSequence encode_sequence(const std::string& s) { return translate(s); }

LocalAligner::LocalAligner(int match, int mismatch, int gap_open, int gap_extend)
    : match_(match), mismatch_(mismatch), gap_open_(gap_open), gap_extend_(gap_extend) {
  for (int a = 0; a < 5; ++a) {
    for (int b = 0; b < 5; ++b) mat_[a * 5 + b] = static_cast<int8_t>(a == b && a < 4 ? match : -mismatch);
  }
}

void LocalAligner::set_reference(const std::string& reference) { ref_ = translate(reference); }

// One pass of the local-alignment recurrence, reference column by reference column:
//   H(i,j) = max(0, H(i-1,j-1) + s(i,j), E(i,j), F(i,j))
//   E(i+1,j) = max(E(i,j) - ge, H(i,j) - go)      gap that consumes reference
//   F(i,j+1) = max(F(i,j) - ge, H(i,j) - go)      gap that consumes query
// Reports the first column (in walking order) whose maximum exceeds every earlier one and
// the smallest query index holding the maximum in it; stops once `stop_at` is reached.
void LocalAligner::sweep(const int8_t* ref, int ref_first, int ref_last, int dir,
                         const std::vector<int8_t>& q, int stop_at, int* best, int* best_ref,
                         int* best_q) const {
  const int n = static_cast<int>(q.size());
  std::vector<int> prev(n, 0), cur(n, 0), e_col(n, 0), best_col;
  *best = 0;
  *best_ref = -1;
  *best_q = -1;
  for (int i = ref_first; i != ref_last + dir; i += dir) {
    int f = 0, diag = 0, col_max = 0;
    const int8_t r = ref[i];
    for (int j = 0; j < n; ++j) {
      int h = diag + score(r, q[j]);
      const int e = e_col[j];
      h = std::max(std::max(h, e), std::max(f, 0));
      diag = prev[j];
      cur[j] = h;
      col_max = std::max(col_max, h);
      const int open = h - gap_open_;
      e_col[j] = std::max(e - gap_extend_, open);
      f = std::max(f - gap_extend_, open);
    }
    prev.swap(cur);
    if (col_max > *best) {
      *best = col_max;
      *best_ref = i;
      best_col = prev;
    }
    if (*best == stop_at) break;
  }
  for (int j = 0; j < n && *best > 0; ++j) {
    if (best_col[j] == *best) {
      *best_q = j;
      break;
    }
  }
}

// Banded re-alignment of ref[0..ref_len) against q[0..q_len) that must reach `target` in
// the bottom-right cell region; cells outside the band count as 0.  Directions:
//   1 diagonal | 2 insertion, extended | 3 insertion, opened | 4 deletion, extended |
//   5 deletion, opened
bool LocalAligner::banded_cigar(const int8_t* ref, int ref_len, const int8_t* q, int q_len, int target,
                                std::vector<std::pair<char, int>>* ops) const {
  int band = std::abs(ref_len - q_len) + 1;
  const size_t cells = static_cast<size_t>(q_len) * ref_len;
  if (cells > (size_t{1} << 27)) return false;   // 128 M cells (~1.5 GB of tables): not a window-sized problem
  // every cell read below was written in the same band iteration (a row's neighbours above
  // and to the left are read only when they lie inside that row's band), so the tables are
  // neither cleared nor re-allocated between calls
  static thread_local std::vector<int> H, E;
  static thread_local std::vector<uint8_t> dE, dF, dH;
  if (H.size() < cells) {
    H.resize(cells);
    E.resize(cells);
    dE.resize(cells);
    dF.resize(cells);
    dH.resize(cells);
  }
  int best = 0;
  for (;;) {
    for (int i = 0; i < q_len; ++i) {
      const int beg = std::max(0, i - band), end = std::min(ref_len - 1, i + band);
      const int up_beg = std::max(0, i - 1 - band), up_end = std::min(ref_len - 1, i - 1 + band);
      int f = 0;
      for (int j = beg; j <= end; ++j) {
        const size_t c = static_cast<size_t>(i) * ref_len + j;
        int t1, t2;
        if (i == 0) {
          t1 = -gap_open_;
          t2 = -gap_extend_;
        } else {
          const bool in_band = j >= up_beg && j <= up_end;
          t1 = (in_band ? H[c - ref_len] : 0) - gap_open_;
          t2 = (in_band ? E[c - ref_len] : 0) - gap_extend_;
        }
        E[c] = std::max(t1, t2);
        dE[c] = t1 > t2 ? 3 : 2;
        t1 = (j > beg ? H[c - 1] : 0) - gap_open_;
        t2 = f - gap_extend_;
        f = std::max(t1, t2);
        dF[c] = t1 > t2 ? 5 : 4;
        const int e1 = std::max(E[c], 0), f1 = std::max(f, 0);
        t1 = std::max(e1, f1);
        t2 = (i > 0 && j > 0 ? H[c - ref_len - 1] : 0) + score(ref[j], q[i]);
        H[c] = std::max(t1, t2);
        best = std::max(best, H[c]);
        dH[c] = t1 <= t2 ? 1 : (e1 > f1 ? dE[c] : dF[c]);
      }
    }
    if (best >= target) break;
    if (band > 2 * (ref_len + q_len)) return false;   // cannot happen for a consistent target
    band *= 2;
  }
  // trace back from the bottom-right cell
  int i = q_len - 1, j = ref_len - 1, state = 2, run = 0;
  char op = 'M', prev_op = 'M';
  std::vector<std::pair<char, int>> rev;
  while (i > 0) {
    if (j < 0 || j < i - band || j > i + band) return false;   // left the band: nothing was computed there
    const size_t c = static_cast<size_t>(i) * ref_len + j;
    const int d = state == 2 ? dH[c] : state == 0 ? dE[c] : dF[c];
    switch (d) {
      case 1: --i; --j; state = 2; op = 'M'; break;
      case 2: --i; state = 0; op = 'I'; break;
      case 3: --i; state = 2; op = 'I'; break;
      case 4: --j; state = 1; op = 'D'; break;
      case 5: --j; state = 2; op = 'D'; break;
      default: return false;
    }
    if (op == prev_op) {
      ++run;
    } else {
      rev.emplace_back(prev_op, run);
      prev_op = op;
      run = 1;
    }
  }
  if (op == 'M') {
    rev.emplace_back('M', run + 1);   // the first cell of the alignment is a match
  } else {
    rev.emplace_back(op, run);
    rev.emplace_back('M', 1);
  }
  ops->clear();
  for (size_t k = rev.size(); k-- > 0;) {
    if (rev[k].second > 0) ops->push_back(rev[k]);
  }
  return true;
}

bool LocalAligner::align(const std::string& query, LocalAlignment* out) const {
  *out = LocalAlignment();
  if (query.empty() || ref_.empty()) return false;
  const std::vector<int8_t> q = translate(query);
  int score1, ref_end, q_end;
  sweep(ref_.data(), 0, static_cast<int>(ref_.size()) - 1, +1, q, -1, &score1, &ref_end, &q_end);
  return finish(ref_, q, score1, ref_end, q_end, out);
}

void LocalAligner::align_to_many(const std::vector<const CodedSequence*>& references, const std::string& query,
                                 std::vector<LocalAlignment>* out, std::vector<char>* ok) const {
  const CodedSequence q = translate(query);
  std::vector<const CodedSequence*> queries(references.size(), &q);
  align_pairs(references, queries, out, ok);
}

void LocalAligner::align_many_to_reference(const std::vector<std::string>& queries, std::vector<LocalAlignment>* out,
                                           std::vector<char>* ok) const {
  std::vector<CodedSequence> coded;
  coded.reserve(queries.size());
  for (const std::string& s : queries) coded.push_back(translate(s));
  std::vector<const CodedSequence*> refs(queries.size(), &ref_), qs;
  for (const CodedSequence& c : coded) qs.push_back(&c);
  align_pairs(refs, qs, out, ok);
}

void LocalAligner::align_pairs(const std::vector<const CodedSequence*>& references,
                               const std::vector<const CodedSequence*>& queries, std::vector<LocalAlignment>* out,
                               std::vector<char>* ok) const {
  const size_t m = references.size();
  out->assign(m, LocalAlignment());
  ok->assign(m, 0);
  const SweepParams params{match_, mismatch_, gap_open_, gap_extend_};
  const int host_width = sweep_width();
  size_t count = 0;
  for (size_t base = 0; base < m; base += count) {
    // a tail of at most 16 pairs goes through the narrow sweep: half the work per column
    const int width = m - base > 16 ? host_width : 16;
    count = std::min<size_t>(width, m - base);
    // int16 lanes: scores and column indices must fit
    bool lanes_fit = count > 1 && gap_open_ < 16000 && mismatch_ < 16000 && match_ < 16000;
    for (size_t l = 0; l < count; ++l) {
      lanes_fit = lanes_fit && static_cast<int64_t>(match_) * static_cast<int64_t>(queries[base + l]->size()) < 32000 &&
                  references[base + l]->size() < 32000;
    }
    if (!lanes_fit) {
      for (size_t l = 0; l < count; ++l) {
        const CodedSequence& ref = *references[base + l];
        const CodedSequence& q = *queries[base + l];
        if (ref.empty() || q.empty()) continue;
        int score1, ref_end, q_end;
        sweep(ref.data(), 0, static_cast<int>(ref.size()) - 1, +1, q, -1, &score1, &ref_end, &q_end);
        (*ok)[base + l] = finish(ref, q, score1, ref_end, q_end, &(*out)[base + l]);
      }
      continue;
    }
    SweepLane fwd[kMaxLanes], rev[kMaxLanes];
    for (size_t l = 0; l < count; ++l) {
      const CodedSequence& ref = *references[base + l];
      const CodedSequence& q = *queries[base + l];
      if (ref.empty() || q.empty()) continue;
      fwd[l].ref = ref.data();
      fwd[l].ref_len = static_cast<int>(ref.size());
      fwd[l].q = q.data();
      fwd[l].q_len = static_cast<int>(q.size());
    }
    sweep_lanes(fwd, params, width);
    bool any_reverse = false;
    for (size_t l = 0; l < count; ++l) {
      if (fwd[l].ref_len == 0) continue;
      (*out)[base + l].score = fwd[l].best;
      if (fwd[l].best <= 0) {
        (*ok)[base + l] = 1;                      // nothing aligned: align() returns true with score 0
        continue;
      }
      rev[l].ref = fwd[l].ref + fwd[l].best_ref;  // walk back from the end point
      rev[l].ref_step = -1;
      rev[l].ref_len = fwd[l].best_ref + 1;
      rev[l].q = fwd[l].q + fwd[l].best_q;
      rev[l].q_step = -1;
      rev[l].q_len = fwd[l].best_q + 1;
      rev[l].stop_at = fwd[l].best;
      any_reverse = true;
    }
    if (!any_reverse) continue;
    sweep_lanes(rev, params, width);
    for (size_t l = 0; l < count; ++l) {
      if (rev[l].ref_len == 0 || rev[l].best != fwd[l].best) continue;      // ok stays 0, as in finish()
      const int ref_end = fwd[l].best_ref, q_end = fwd[l].best_q;
      (*ok)[base + l] = describe(*references[base + l], *queries[base + l], fwd[l].best, ref_end - rev[l].best_ref,
                                 ref_end, q_end - rev[l].best_q, q_end, &(*out)[base + l]);
    }
  }
}

bool LocalAligner::finish(const CodedSequence& ref, const CodedSequence& q, int score1, int ref_end, int q_end,
                          LocalAlignment* out) const {
  *out = LocalAlignment();
  out->score = score1;
  if (score1 <= 0) return true;
  std::vector<int8_t> rq(q.begin(), q.begin() + q_end + 1);
  std::reverse(rq.begin(), rq.end());
  int score2, ref_begin, k;
  sweep(ref.data(), ref_end, 0, -1, rq, score1, &score2, &ref_begin, &k);
  if (score2 != score1) return false;
  return describe(ref, q, score1, ref_begin, ref_end, q_end - k, q_end, out);
}

// CIGAR and text form of the alignment whose corner points are known
bool LocalAligner::describe(const CodedSequence& ref, const CodedSequence& q, int score1, int ref_begin, int ref_end,
                            int q_begin, int q_end, LocalAlignment* out) const {
  *out = LocalAlignment();
  out->score = score1;
  const int q_len = static_cast<int>(q.size());
  std::vector<std::pair<char, int>> ops;
  if (!banded_cigar(ref.data() + ref_begin, ref_end - ref_begin + 1, q.data() + q_begin,
                    q_end - q_begin + 1, score1, &ops)) {
    return false;
  }
  out->ref_begin = ref_begin;
  out->ref_end = ref_end;
  out->query_begin = q_begin;
  out->query_end = q_end;
  // text form: soft clips around, M runs split into '=' and 'X'
  std::string cigar;
  auto emit = [&](int len, char c) {
    if (len > 0) cigar += std::to_string(len) + c;
  };
  emit(q_begin, 'S');
  int ri = ref_begin, qi = q_begin;
  for (const auto& o : ops) {
    if (o.first == 'M') {
      int run = 0;
      bool run_eq = true;
      for (int t = 0; t < o.second; ++t, ++ri, ++qi) {
        const bool eq = ref[ri] == q[qi];
        if (run && eq != run_eq) {
          emit(run, run_eq ? '=' : 'X');
          run = 0;
        }
        run_eq = eq;
        ++run;
        out->mismatches += eq ? 0 : 1;
      }
      emit(run, run_eq ? '=' : 'X');
    } else if (o.first == 'I') {
      emit(o.second, 'I');
      qi += o.second;
      out->mismatches += o.second;
    } else {
      emit(o.second, 'D');
      ri += o.second;
      out->mismatches += o.second;
    }
  }
  emit(q_len - q_end - 1, 'S');
  out->cigar = cigar;
  return true;
}

}  // namespace dv
