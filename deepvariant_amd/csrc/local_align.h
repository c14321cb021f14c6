// Local (Smith-Waterman, affine gaps) alignment with the observable behaviour of the
// library the reference links for this: Complete-Striped-Smith-Waterman-Library v1.2.5
// (WORKSPACE:32-40, third_party/libssw.BUILD; its sources are NOT in the reference tree),
// as driven by deepvariant/realigner/ssw.{h,cc} and fast_pass_aligner.cc:161-180.
//
// What is restated (from the library's published algorithm, Zhao et al. 2013, and its
// documented interface) is everything that decides WHICH optimal alignment is reported:
//   * score matrix of StripedSmithWaterman::Aligner: +match on A/C/G/T identity,
//     -mismatch otherwise, N (any other letter) scores -mismatch against everything;
//   * a gap of length g costs gap_open + (g - 1) * gap_extend;
//   * end point: the FIRST reference column reaching the maximum score and, in it, the
//     smallest query index holding that score;
//   * begin point: the same search on the reversed prefixes, stopped at the first column
//     that reaches the forward score;
//   * CIGAR: banded global re-alignment of the two sub-sequences, band |dr - dq| + 1 and
//     doubling until the score is reached, traced back with the priorities
//     diagonal >= gap, gap-extension >= gap-open, deletion >= insertion on ties;
//   * M runs split into '=' / 'X', query overhangs as 'S'.
// The SIMD striping of the library is an implementation detail (its H values equal the
// scalar recurrence's), so this is plain scalar code: the aligner runs on the host, once
// per (candidate, alt allele, read).
#ifndef DV_LOCAL_ALIGN_H_
#define DV_LOCAL_ALIGN_H_

#include <cstdint>
#include <string>
#include <vector>

namespace dv {

struct LocalAlignment {
  int score = 0;
  int ref_begin = -1, ref_end = -1;      // inclusive, 0-based
  int query_begin = -1, query_end = -1;  // inclusive, 0-based
  int mismatches = 0;
  std::string cigar;                     // e.g. "3S4=1X4=1I5=2S"; empty if nothing aligned
};

// a reference sequence in the aligner's 0..4 codes (A C G T other)
using CodedSequence = std::vector<int8_t>;
CodedSequence encode_sequence(const std::string& s);

class LocalAligner {
 public:
  LocalAligner(int match, int mismatch, int gap_open, int gap_extend);
  void set_reference(const std::string& reference);
  // false if the query or the reference is empty (libssw's Align fails the same way), or if
  // the aligned sub-problem exceeds 128 M cells (reads x haplotype here are window-sized)
  bool align(const std::string& query, LocalAlignment* out) const;
  // The same alignments of ONE query against MANY references (the realigner aligns a read to
  // every assembled haplotype): the forward pass -- nine tenths of the work -- runs for 16
  // references at a time, one per int16 SIMD lane, each lane the exact scalar recurrence.
  // ok[k] is what align() would return for references[k].
  void align_to_many(const std::vector<const CodedSequence*>& references, const std::string& query,
                     std::vector<LocalAlignment>* out, std::vector<char>* ok) const;
  // ... MANY queries against the reference of set_reference() (haplotypes -> reference window)
  void align_many_to_reference(const std::vector<std::string>& queries, std::vector<LocalAlignment>* out,
                               std::vector<char>* ok) const;
  // ... and the general form: pair k = (references[k], queries[k]); both passes of the search
  // (end point, then start point on the reversed prefixes) run 16 pairs at a time
  void align_pairs(const std::vector<const CodedSequence*>& references,
                   const std::vector<const CodedSequence*>& queries, std::vector<LocalAlignment>* out,
                   std::vector<char>* ok) const;

 private:
  int score(int8_t a, int8_t b) const { return mat_[a * 5 + b]; }
  // best local score over ref[r0..r1] (walked in direction dir) x q; see .cpp
  void sweep(const int8_t* ref, int ref_first, int ref_last, int dir, const std::vector<int8_t>& q,
             int stop_at, int* best, int* best_ref, int* best_q) const;
  bool banded_cigar(const int8_t* ref, int ref_len, const int8_t* q, int q_len, int target,
                    std::vector<std::pair<char, int>>* ops) const;
  // reverse pass + CIGAR for a forward result (score1 at ref_end / q_end)
  bool finish(const CodedSequence& ref, const CodedSequence& q, int score1, int ref_end, int q_end,
              LocalAlignment* out) const;
  bool describe(const CodedSequence& ref, const CodedSequence& q, int score1, int ref_begin, int ref_end,
                int q_begin, int q_end, LocalAlignment* out) const;

  int match_, mismatch_, gap_open_, gap_extend_;
  int8_t mat_[25];
  std::vector<int8_t> ref_;
};

}  // namespace dv

#endif  // DV_LOCAL_ALIGN_H_
