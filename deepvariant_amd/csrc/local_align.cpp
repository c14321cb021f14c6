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

}  // namespace

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
  std::vector<int> H(cells), E(cells);
  std::vector<uint8_t> dE(cells), dF(cells), dH(cells);
  int best = 0;
  for (;;) {
    std::fill(H.begin(), H.end(), 0);
    std::fill(E.begin(), E.end(), 0);
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
    if (j < 0) return false;
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
  const int ref_len = static_cast<int>(ref_.size()), q_len = static_cast<int>(q.size());
  int score1, ref_end, q_end;
  sweep(ref_.data(), 0, ref_len - 1, +1, q, -1, &score1, &ref_end, &q_end);
  out->score = score1;
  if (score1 <= 0) return true;
  std::vector<int8_t> rq(q.begin(), q.begin() + q_end + 1);
  std::reverse(rq.begin(), rq.end());
  int score2, ref_begin, k;
  sweep(ref_.data(), ref_end, 0, -1, rq, score1, &score2, &ref_begin, &k);
  if (score2 != score1) return false;
  const int q_begin = q_end - k;
  std::vector<std::pair<char, int>> ops;
  if (!banded_cigar(ref_.data() + ref_begin, ref_end - ref_begin + 1, q.data() + q_begin,
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
        const bool eq = ref_[ri] == q[qi];
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
