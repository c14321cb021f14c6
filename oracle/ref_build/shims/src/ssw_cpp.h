// TEST INFRASTRUCTURE (oracle/ref_build): the C++ interface of the Complete-Striped-Smith-Waterman-Library
// (src/ssw_cpp.h of libssw v1.2.5, which the reference links but does not vendor: WORKSPACE:32-40) for the
// reference's deepvariant/realigner/ssw.{h,cc} and fast_pass_aligner.cc -- implemented on the PRODUCT's restatement
// of that library (deepvariant_amd/csrc/local_align.cpp, pinned by the vectors of ssw_test.cc / ssw_wrap_test.py).
// What this build checks is therefore the reference's FastPassAligner / alt_aligned_pileup_lib code against the
// product's restatements of THOSE, with one and the same local aligner underneath both.
#ifndef DVREF_SSW_CPP_SHIM_H_
#define DVREF_SSW_CPP_SHIM_H_
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
#include "local_align.h"
namespace StripedSmithWaterman {
struct Alignment {
  uint16_t sw_score = 0;
  uint16_t sw_score_next_best = 0;
  int32_t ref_begin = 0, ref_end = 0, query_begin = 0, query_end = 0, ref_end_next_best = 0, mismatches = 0;
  std::string cigar_string;
  std::vector<uint32_t> cigar;
  void Clear() { *this = Alignment(); }
};
struct Filter {
  bool report_begin_position = true, report_cigar = true;
  uint16_t score_filter = 0, distance_filter = 32767;
  Filter() = default;
  Filter(const bool& pos, const bool& cigar, const uint16_t& score, const uint16_t& dis)
      : report_begin_position(pos), report_cigar(cigar), score_filter(score), distance_filter(dis) {}
};
class Aligner {
 public:
  Aligner() : Aligner(2, 2, 3, 1) {}
  Aligner(const uint8_t& match, const uint8_t& mismatch, const uint8_t& gap_open, const uint8_t& gap_extend)
      : impl_(new dv::LocalAligner(match, mismatch, gap_open, gap_extend)) {}
  int SetReferenceSequence(const char* seq, const int& length) {
    impl_->set_reference(std::string(seq, static_cast<size_t>(length)));
    return length;
  }
  // libssw v1.2.5 returns the s_align flag word: 0 = no error (fast_pass_aligner.cc:193 tests `== 0`).  Its early
  // exits (no reference, empty query) `return false`, i.e. 0 as well, leaving the alignment cleared (sw_score 0).
  uint16_t Align(const char* query, const Filter& filter, Alignment* alignment, const int32_t /*maskLen*/) const {
    dv::LocalAlignment a;
    alignment->Clear();
    (void)filter;
    if (!impl_->align(query, &a)) return 0;
    alignment->sw_score = static_cast<uint16_t>(a.score);
    alignment->ref_begin = a.ref_begin;
    alignment->ref_end = a.ref_end;
    alignment->query_begin = a.query_begin;
    alignment->query_end = a.query_end;
    alignment->mismatches = a.mismatches;
    alignment->cigar_string = a.cigar;
    return 0;
  }
 private:
  std::shared_ptr<dv::LocalAligner> impl_;
};
}  // namespace StripedSmithWaterman
#endif
