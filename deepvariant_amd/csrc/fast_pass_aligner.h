// Read -> haplotype -> reference realignment: the host-side aligner behind alt-aligned
// pileups (RealignReadsToHaplotype, deepvariant/alt_aligned_pileup_lib.cc:278-313) and the
// window realigner (deepvariant/realigner/realigner.py).  Behaviour of
// deepvariant/realigner/fast_pass_aligner.{h,cc}; tests mirror fast_pass_aligner_test.cc.
//
//   1. k-mer index over the reads; every haplotype position looks its k-mer up and tries
//      the read at the implied offset, accepting <= max_num_of_mismatches substitutions
//      (no indels) -- the "fast pass";
//   2. haplotypes are aligned to the reference window (LocalAligner), unless identical;
//   3. reads the fast pass left without any alignment go through LocalAligner against
//      every supported haplotype (only the reference haplotype when force_alignment);
//   4. per read the best haplotype wins (ties: a non-reference haplotype) and the
//      read->haplotype and haplotype->reference CIGARs are merged base by base.
#ifndef DV_FAST_PASS_ALIGNER_H_
#define DV_FAST_PASS_ALIGNER_H_

#include <cstdint>
#include <memory>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "local_align.h"

namespace dv {

// nucleus CigarUnit::Operation values used here (third_party/nucleus/protos/cigar.proto)
enum CigarOpKind : int { kOpUnspecified = 0, kOpMatch = 1, kOpInsert = 2, kOpDelete = 3, kOpSoftClip = 5 };

struct CigarOp {
  int op = kOpUnspecified;
  int length = 0;
  bool operator==(const CigarOp& o) const { return op == o.op && length == o.length; }
};
using Cigar = std::vector<CigarOp>;

Cigar parse_cigar(std::string_view text);          // "(\d+)([XIDS=])" tokens; '=' and 'X' -> match
std::string cigar_text(const Cigar& cigar);        // M / I / D / S
void merge_cigar_op(const CigarOp& op, int read_len, Cigar* cigar);
std::vector<int> positions_map(std::string_view haplotype_cigar, size_t haplotype_size);

struct ReadAlignment {
  static constexpr uint16_t kNotAligned = 0xffff;
  uint16_t position = kNotAligned;
  std::string cigar;
  int score = 0;
  void reset() { *this = ReadAlignment(); }
};

struct HaplotypeAlignment {
  size_t haplotype_index = 0;
  int haplotype_score = 0;
  std::vector<ReadAlignment> reads;
  std::string cigar;
  Cigar cigar_ops;
  uint64_t ref_pos = 0;
  std::vector<int> hap_to_ref;
  bool is_reference = false;
};

struct AlignerOptions {     // 0 = keep the class default, as the reference's proto
  int match = 0, mismatch = 0, gap_open = 0, gap_extend = 0;
  int kmer_size = 0, read_size = 0, max_num_of_mismatches = 0;
  double similarity_threshold = 0.0;
  bool force_alignment = false;
};

struct RealignedRead {
  int status = 0;            // 0 original alignment kept | 1 new alignment | 2 dropped (force_alignment)
  int64_t position = 0;
  Cigar cigar;
};

class FastPassAligner {
 public:
  void set_reference(const std::string& r) { reference_ = r; }
  void set_reads(const std::vector<std::string>& r) { reads_ = r; }
  void set_ref_start(uint64_t position) { region_position_ = position; }
  void set_haplotypes(const std::vector<std::string>& h) { haplotypes_ = h; }
  void set_normalize_reads(bool v) { normalize_reads_ = v; }
  void set_ref_prefix_len(int v) { ref_prefix_len_ = v; }
  void set_ref_suffix_len(int v) { ref_suffix_len_ = v; }
  bool set_options(const AlignerOptions& o, std::string* error);

  std::vector<RealignedRead> align_reads(const std::vector<std::string>& sequences);

  void build_index();
  void fast_align_reads_to_haplotype(std::string_view haplotype, int* haplotype_score,
                                     std::vector<ReadAlignment>* alignments) const;
  void init_local_aligner();
  void align_haplotypes_to_reference();
  void calculate_position_maps();
  void local_align_reads_to_haplotypes(int score_threshold);
  void calculate_score_threshold();
  bool best_read_alignment(size_t read, int* best_hap) const;
  bool calculate_read_to_ref_alignment(size_t read_index, const ReadAlignment& read_to_hap,
                                       const Cigar& hap_to_ref, Cigar* out, std::string* error) const;
  bool is_alignment_normalized(const Cigar& cigar, int ref_offset, std::string_view read) const;

  const std::vector<HaplotypeAlignment>& haplotype_alignments() const { return alignments_; }
  const std::vector<std::string>& reads() const { return reads_; }
  int score_threshold() const { return score_threshold_; }
  size_t index_size() const { return index_.size(); }
  // occurrences of a k-mer: (read, offset) pairs in insertion order
  std::vector<std::pair<uint32_t, uint32_t>> kmer_occurrences(std::string_view kmer) const;

 private:
  void fast_align_reads_to_haplotypes();
  int fast_align_strings(std::string_view a, std::string_view b, int max_mismatches, int* mismatches) const;

  std::string reference_;
  uint64_t region_position_ = 0;
  std::vector<std::string> haplotypes_;
  std::vector<HaplotypeAlignment> alignments_;
  std::unordered_map<std::string_view, std::vector<std::pair<uint32_t, uint32_t>>> index_;
  std::vector<std::string> reads_;
  int kmer_size_ = 32;
  int read_size_ = 100;
  int max_num_of_mismatches_ = 2;
  int16_t score_threshold_ = 0;
  int match_ = 4, mismatch_ = 6, gap_open_ = 8, gap_extend_ = 1;
  bool force_alignment_ = false;
  double similarity_threshold_ = 0.85;
  std::unique_ptr<LocalAligner> aligner_;
  int ref_prefix_len_ = 0, ref_suffix_len_ = 0;
  bool normalize_reads_ = false;
};

}  // namespace dv

#endif  // DV_FAST_PASS_ALIGNER_H_
