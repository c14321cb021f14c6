// See fast_pass_aligner.h.  Section comments cite deepvariant/realigner/fast_pass_aligner.cc.
#include "fast_pass_aligner.h"

#include <algorithm>
#include <cctype>
#include <deque>

namespace dv {

// ---------------------------------------------------------------- CIGAR helpers
Cigar parse_cigar(std::string_view text) {   // CigarStringToVector, :316-329
  Cigar out;
  size_t i = 0;
  while (i < text.size()) {
    size_t j = i;
    int len = 0;
    while (j < text.size() && std::isdigit(static_cast<unsigned char>(text[j]))) {
      len = len * 10 + (text[j] - '0');
      ++j;
    }
    if (j == i || j >= text.size()) break;
    int op;
    switch (text[j]) {
      case '=': case 'X': case 'M': op = kOpMatch; break;   // 'M': this library's own text form
      case 'S': op = kOpSoftClip; break;
      case 'D': op = kOpDelete; break;
      case 'I': op = kOpInsert; break;
      default: return out;                   // the pattern stops matching
    }
    out.push_back({op, len});
    i = j + 1;
  }
  return out;
}

std::string cigar_text(const Cigar& cigar) {
  std::string s;
  for (const CigarOp& o : cigar) {
    s += std::to_string(o.length);
    s += o.op == kOpMatch ? 'M' : o.op == kOpInsert ? 'I' : o.op == kOpDelete ? 'D' : o.op == kOpSoftClip ? 'S' : '?';
  }
  return s;
}

static int aligned_length(const Cigar& cigar) {   // AlignedLength, :677-685
  int len = 0;
  for (const CigarOp& o : cigar) len += o.op != kOpDelete ? o.length : 0;
  return len;
}

// MergeCigarOp, :697-757.  An insertion arriving right after a deletion (or the reverse)
// cancels one base of it into a match placed in front of the trailing indel.
void merge_cigar_op(const CigarOp& op, int read_len, Cigar* cigar) {
  const int last_op = cigar->empty() ? kOpUnspecified : cigar->back().op;
  const int before = aligned_length(*cigar);
  const int new_len = op.op != kOpDelete ? std::min(op.length, read_len - before) : op.length;
  if (new_len <= 0 || before == read_len) return;
  if ((op.op == kOpInsert && last_op == kOpDelete) || (op.op == kOpDelete && last_op == kOpInsert)) {
    const size_t last = cigar->size() - 1;
    const size_t before_last = cigar->size() > 1 ? cigar->size() - 2 : last;
    if ((*cigar)[before_last].op != kOpMatch) {
      cigar->insert(cigar->begin() + last, CigarOp{kOpMatch, 1});
    } else {
      (*cigar)[before_last].length += 1;
    }
    if (cigar->back().length == 1) {
      cigar->pop_back();
    } else {
      cigar->back().length -= 1;
    }
  } else if (op.op == last_op) {
    cigar->back().length += new_len;
  } else {
    cigar->push_back({op.op, new_len});
  }
}

// SetPositionsMap, :601-650: for every haplotype position the shift that turns
// (haplotype offset) into (reference offset).
std::vector<int> positions_map(std::string_view cigar, size_t haplotype_size) {
  std::vector<int> map(haplotype_size, 0);
  int shift = 0;
  size_t pos = 0, i = 0;
  auto put = [&](int v) {
    if (pos < map.size()) map[pos] = v;
    ++pos;
  };
  while (i < cigar.size()) {
    size_t j = i;
    int len = 0;
    while (j < cigar.size() && std::isdigit(static_cast<unsigned char>(cigar[j]))) {
      len = len * 10 + (cigar[j] - '0');
      ++j;
    }
    if (j == i || j >= cigar.size()) break;
    switch (cigar[j]) {
      case '=': case 'X':
        for (int k = 0; k < len; ++k) put(shift);
        break;
      case 'S':
        shift -= len;
        for (int k = 0; k < len; ++k) put(shift);
        break;
      case 'D':
        shift += len;
        break;
      case 'I':
        for (int k = 0; k < len; ++k) {
          put(shift);
          --shift;
        }
        break;
      default:
        return map;
    }
    i = j + 1;
  }
  return map;
}

// ---------------------------------------------------------------- options
bool FastPassAligner::set_options(const AlignerOptions& o, std::string* error) {   // :79-113
  if (o.kmer_size > 0) kmer_size_ = o.kmer_size;
  if (o.read_size > 0) read_size_ = o.read_size;
  if (o.max_num_of_mismatches > 0) max_num_of_mismatches_ = o.max_num_of_mismatches;
  if (o.similarity_threshold > 0.0) similarity_threshold_ = o.similarity_threshold;
  if (o.match > 0) match_ = o.match;
  if (o.mismatch > 0) mismatch_ = o.mismatch;
  if (o.gap_open > 0) gap_open_ = o.gap_open;
  if (o.gap_extend > 0) gap_extend_ = o.gap_extend;
  force_alignment_ = o.force_alignment;
  if (kmer_size_ < 3 || kmer_size_ > 32) {
    *error = "Check failed: kmer_size_ >= 3 && kmer_size_ <= 32";
    return false;
  }
  if (similarity_threshold_ < 0.0 || similarity_threshold_ > 1.0) {
    *error = "Check failed: similarity_threshold_ in [0, 1]";
    return false;
  }
  return true;
}

void FastPassAligner::calculate_score_threshold() {   // :115-125
  const double t = static_cast<double>(match_) * read_size_ * similarity_threshold_ -
                   static_cast<double>(mismatch_) * read_size_ * (1 - similarity_threshold_);
  score_threshold_ = static_cast<int16_t>(t);
  if (score_threshold_ < 0) score_threshold_ = 1;
}

// ---------------------------------------------------------------- index + fast pass
void FastPassAligner::build_index() {   // :586-599
  index_.clear();
  for (uint32_t r = 0; r < reads_.size(); ++r) {
    const std::string& read = reads_[r];
    if (static_cast<int>(read.size()) <= kmer_size_) continue;   // left to the local aligner
    const std::string_view view(read);
    for (uint32_t i = 0; i + kmer_size_ <= read.size(); ++i) {
      index_[view.substr(i, kmer_size_)].emplace_back(r, i);
    }
  }
}

std::vector<std::pair<uint32_t, uint32_t>> FastPassAligner::kmer_occurrences(std::string_view kmer) const {
  auto it = index_.find(kmer);
  return it == index_.end() ? std::vector<std::pair<uint32_t, uint32_t>>() : it->second;
}

int FastPassAligner::fast_align_strings(std::string_view a, std::string_view b, int max_mismatches,
                                        int* mismatches) const {   // :289-309
  int matches = 0;
  *mismatches = 0;
  for (size_t i = 0; i < a.size(); ++i) {
    if (a[i] != b[i] && a[i] != 'N' && b[i] != 'N') {
      if (++*mismatches == max_mismatches) return 0;
    } else {
      ++matches;
    }
  }
  return matches * match_ - *mismatches * mismatch_;
}

void FastPassAligner::fast_align_reads_to_haplotype(std::string_view haplotype, int* haplotype_score,
                                                    std::vector<ReadAlignment>* alignments) const {   // :207-286
  const bool is_ref = haplotype == reference_;
  std::vector<int> coverage(haplotype.size(), 0);
  const int last_pos = static_cast<int>(haplotype.size()) - kmer_size_;
  for (int i = 0; i <= last_pos; ++i) {
    auto hit = index_.find(haplotype.substr(i, kmer_size_));
    if (hit == index_.end()) continue;
    for (const auto& occ : hit->second) {
      const size_t read_id = occ.first;
      const size_t start = static_cast<size_t>(std::max<int64_t>(0, static_cast<int64_t>(i) - occ.second));
      const size_t span = reads_[read_id].size();
      if (start + span > haplotype.size()) continue;
      ReadAlignment& ra = (*alignments)[read_id];
      if (ra.position != ReadAlignment::kNotAligned && ra.position == start) continue;
      int mismatches = 0;
      const int score = fast_align_strings(haplotype.substr(start, span), reads_[read_id],
                                           max_num_of_mismatches_ + 1, &mismatches);
      if (mismatches <= max_num_of_mismatches_) {
        const int old_score = ra.score;
        for (size_t p = start; p < start + span; ++p) ++coverage[p];
        if (old_score < score) {
          ra.score = score;
          *haplotype_score += score - old_score;
          ra.position = static_cast<uint16_t>(start);
          ra.cigar = std::to_string(span) + "=";
        }
      }
    }
    // a haplotype position no read supports discards the haplotype, except inside the
    // reference padding and for the reference haplotype itself
    if (coverage[i] == 0 && i >= ref_prefix_len_ &&
        static_cast<size_t>(i) < haplotype.size() - static_cast<size_t>(ref_suffix_len_) && !is_ref) {
      *haplotype_score = 0;
      return;
    }
  }
}

void FastPassAligner::fast_align_reads_to_haplotypes() {   // :182-205
  std::vector<ReadAlignment> scratch(reads_.size());
  for (size_t i = 0; i < haplotypes_.size(); ++i) {
    int score = 0;
    for (ReadAlignment& ra : scratch) ra.reset();
    fast_align_reads_to_haplotype(haplotypes_[i], &score, &scratch);
    if (score == 0) {
      for (ReadAlignment& ra : scratch) ra.reset();
    }
    HaplotypeAlignment ha;
    ha.haplotype_index = i;
    ha.haplotype_score = score;
    ha.reads = scratch;
    alignments_.push_back(std::move(ha));
  }
}

// ---------------------------------------------------------------- local alignment stages
void FastPassAligner::init_local_aligner() {   // InitSswLib, :155-160
  aligner_ = std::make_unique<LocalAligner>(match_, mismatch_, gap_open_, gap_extend_);
}

void FastPassAligner::align_haplotypes_to_reference() {   // :336-375
  aligner_->set_reference(reference_);
  if (alignments_.empty()) {
    for (size_t i = 0; i < haplotypes_.size(); ++i) {
      HaplotypeAlignment ha;
      ha.haplotype_index = i;
      ha.haplotype_score = -1;
      ha.reads.assign(reads_.size(), ReadAlignment());
      alignments_.push_back(std::move(ha));
    }
  }
  // the haplotypes that differ from the reference are aligned to it together (16 per SIMD batch)
  std::vector<HaplotypeAlignment*> todo;
  std::vector<std::string> queries;
  for (HaplotypeAlignment& ha : alignments_) {
    const std::string& hap = haplotypes_[ha.haplotype_index];
    if (hap == reference_) {
      ha.is_reference = true;
      ha.cigar = std::to_string(hap.size()) + "=";
      ha.cigar_ops = parse_cigar(ha.cigar);
      ha.ref_pos = 0;
    } else {
      todo.push_back(&ha);
      queries.push_back(hap);
    }
  }
  std::vector<LocalAlignment> results;
  std::vector<char> ok;
  aligner_->align_many_to_reference(queries, &results, &ok);
  for (size_t k = 0; k < todo.size(); ++k) {
    const LocalAlignment& al = results[k];
    if (!ok[k] || al.score <= 0) continue;
    HaplotypeAlignment& ha = *todo[k];
    ha.is_reference = al.cigar == std::to_string(queries[k].size()) + "=";
    ha.cigar = al.cigar;
    ha.cigar_ops = parse_cigar(al.cigar);
    ha.ref_pos = static_cast<uint64_t>(al.ref_begin);
  }
}

void FastPassAligner::calculate_position_maps() {   // :652-657
  for (HaplotypeAlignment& ha : alignments_) {
    ha.hap_to_ref = positions_map(ha.cigar, haplotypes_[ha.haplotype_index].size());
  }
}

void FastPassAligner::local_align_reads_to_haplotypes(int score_threshold) {   // :377-418
  const int threshold = static_cast<uint16_t>(score_threshold);
  // the haplotypes a read is aligned to do not depend on the read: encode them once
  std::vector<HaplotypeAlignment*> targets;
  std::vector<CodedSequence> coded;
  for (HaplotypeAlignment& ha : alignments_) {
    const bool forced = force_alignment_ && ha.is_reference;
    if (ha.haplotype_score == 0 && !forced) continue;
    targets.push_back(&ha);
    coded.push_back(encode_sequence(haplotypes_[ha.haplotype_index]));
  }
  // every (unplaced read, haplotype) pair goes through the aligner in one batch call: its SIMD
  // lanes take 16 pairs at a time whatever read or haplotype they belong to
  std::vector<size_t> unplaced;
  std::vector<CodedSequence> coded_reads;
  for (size_t r = 0; r < reads_.size(); ++r) {
    bool aligned = false;
    for (const HaplotypeAlignment& ha : alignments_) aligned = aligned || ha.reads[r].score > 0;
    if (aligned || targets.empty()) continue;
    unplaced.push_back(r);
    coded_reads.push_back(encode_sequence(reads_[r]));
  }
  std::vector<const CodedSequence*> refs, queries;
  for (size_t u = 0; u < unplaced.size(); ++u) {
    for (const CodedSequence& c : coded) {
      refs.push_back(&c);
      queries.push_back(&coded_reads[u]);
    }
  }
  std::vector<LocalAlignment> results;
  std::vector<char> ok;
  aligner_->align_pairs(refs, queries, &results, &ok);
  for (size_t u = 0; u < unplaced.size(); ++u) {
    const size_t r = unplaced[u];
    for (size_t t = 0; t < targets.size(); ++t) {
      HaplotypeAlignment& ha = *targets[t];
      const size_t k = u * targets.size() + t;
      const LocalAlignment& al = results[k];
      if (!ok[k] || al.score <= 0) continue;
      if (al.score >= threshold || (force_alignment_ && ha.is_reference)) {
        ha.reads[r].score = al.score;
        ha.reads[r].cigar = al.cigar;
        ha.reads[r].position = static_cast<uint16_t>(al.ref_begin);
      }
    }
  }
}

bool FastPassAligner::best_read_alignment(size_t read, int* best_hap) const {   // :659-674
  int best = 0;
  bool found = false;
  for (size_t h = 0; h < haplotypes_.size() && h < alignments_.size(); ++h) {
    const int s = alignments_[h].reads[read].score;
    if (s > best || (best > 0 && s == best && !alignments_[h].is_reference)) {
      best = s;
      *best_hap = static_cast<int>(h);
      found = true;
    }
  }
  return found;
}

// ---------------------------------------------------------------- CIGAR merging
// CalculateReadToRefAlignment, :841-968 (+ LeftTrimHaplotypeToRefAlignment :762-795 and
// MergeOneBaseOperations :811-839): walks the read->haplotype and haplotype->reference
// CIGARs one base at a time.
bool FastPassAligner::calculate_read_to_ref_alignment(size_t read_index, const ReadAlignment& read_to_hap,
                                                      const Cigar& hap_to_ref_in, Cigar* out,
                                                      std::string* error) const {
  if (read_index >= reads_.size()) {
    *error = "Check failed: read_index < reads_.size()";
    return false;
  }
  const int read_len = static_cast<int>(reads_[read_index].size());
  const int hap_pos = read_to_hap.position;
  const Cigar r2h_v = parse_cigar(read_to_hap.cigar);
  std::deque<CigarOp> r2h(r2h_v.begin(), r2h_v.end());
  std::deque<CigarOp> h2r(hap_to_ref_in.begin(), hap_to_ref_in.end());
  // drop the part of the haplotype alignment in front of the read
  int cur = 0;
  while (cur != hap_pos) {
    if (h2r.empty()) {
      *error = "Check failed: !haplotype_to_ref_cigar_ops.empty()";
      return false;
    }
    const CigarOp op = h2r.front();
    h2r.pop_front();
    if (op.op == kOpMatch || op.op == kOpSoftClip || op.op == kOpInsert) {
      if (op.length + cur > hap_pos) h2r.push_front({op.op, op.length - (hap_pos - cur)});
      cur = std::min(op.length + cur, hap_pos);
    }
  }
  if (!h2r.empty() && h2r.front().op == kOpDelete) h2r.pop_front();   // cannot start with a deletion
  if (h2r.empty()) {
    *error = "Check failed: !haplotype_to_ref_cigar_ops.empty()";
    return false;
  }
  if (!r2h.empty() && r2h.front().op == kOpSoftClip) {
    merge_cigar_op({kOpSoftClip, r2h.front().length}, read_len, out);
    r2h.pop_front();
  }
  auto merge_one = [&](const CigarOp& a, const CigarOp& b) {
    for (int op : {kOpSoftClip, kOpDelete, kOpInsert, kOpMatch}) {
      if (a.op == op || b.op == op) {
        merge_cigar_op({op, 1}, read_len, out);
        break;
      }
    }
  };
  CigarOp cr, ch;
  while ((!r2h.empty() || !h2r.empty()) && aligned_length(*out) < read_len) {
    if (!r2h.empty() && h2r.empty() && ch.length == 0) {   // soft-clipped tail past the haplotype
      merge_cigar_op(r2h.front(), read_len, out);
      r2h.pop_front();
      continue;
    }
    if (r2h.empty() && cr.length == 0 && !h2r.empty()) break;   // the read is used up
    if (cr.length == 0) {
      cr = r2h.front();
      r2h.pop_front();
    }
    if (ch.length == 0) {
      if (h2r.empty()) break;
      ch = h2r.front();
      h2r.pop_front();
    }
    while (cr.length > 0 && ch.length > 0) {
      if ((cr.op == kOpDelete && ch.op == kOpInsert) || (cr.op == kOpInsert && ch.op == kOpDelete)) {
        --ch.length;
        --cr.length;
        if (ch.op == kOpDelete) {   // a read insertion filling a haplotype deletion is a match
          h2r.push_front({kOpMatch, 1});
          r2h.push_front({kOpMatch, 1});
        }
        continue;
      }
      merge_one(cr, ch);
      if (cr.op == kOpInsert) {
        --cr.length;
      } else if (ch.op == kOpDelete) {
        --ch.length;
      } else {
        --ch.length;
        --cr.length;
      }
    }
  }
  if (cr.length > 0 && cr.op == kOpSoftClip) {
    while (cr.length > 0) {
      merge_one(cr, ch);
      --cr.length;
    }
  }
  if (!r2h.empty() || cr.length > 0) out->clear();   // the read runs past the haplotype
  return true;
}

// IsAlignmentNormalized, :440-484: an indel whose last base equals the base in front of it
// could be shifted left.
bool FastPassAligner::is_alignment_normalized(const Cigar& cigar, int ref_offset, std::string_view read) const {
  if (ref_offset < 0) return true;
  size_t ref_i = static_cast<size_t>(ref_offset), read_i = 0;
  for (const CigarOp& op : cigar) {
    if (op.op == kOpSoftClip) {
      read_i += op.length;
      continue;
    }
    if (op.op != kOpMatch) {
      char last;
      if (op.op == kOpDelete) {
        if (ref_i + op.length > reference_.size()) return false;
        last = op.length > 0 ? reference_[ref_i + op.length - 1] : '\0';
      } else {
        if (read_i + op.length > read.size()) return false;   // CHECK in the reference
        last = op.length > 0 ? read[read_i + op.length - 1] : '\0';
      }
      if ((ref_i > 0 && op.op == kOpInsert && ref_i - 1 < reference_.size() && last == reference_[ref_i - 1]) ||
          (read_i > 0 && op.op == kOpDelete && read_i - 1 < read.size() && last == read[read_i - 1])) {
        return false;
      }
    }
    if (op.op != kOpInsert) ref_i += op.length;
    if (op.op != kOpDelete) read_i += op.length;
  }
  return true;
}

// ---------------------------------------------------------------- entry point
std::vector<RealignedRead> FastPassAligner::align_reads(const std::vector<std::string>& sequences) {   // :131-177
  for (const std::string& s : sequences) {
    std::string up(s);
    for (char& c : up) c = static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
    reads_.push_back(std::move(up));
  }
  calculate_score_threshold();
  build_index();
  fast_align_reads_to_haplotypes();
  init_local_aligner();
  align_haplotypes_to_reference();
  calculate_position_maps();
  local_align_reads_to_haplotypes(score_threshold_);
  std::sort(alignments_.begin(), alignments_.end(),
            [](const HaplotypeAlignment& a, const HaplotypeAlignment& b) {
              return a.haplotype_score < b.haplotype_score;
            });
  // RealignReadsToReference, :486-571
  std::vector<RealignedRead> out(sequences.size());
  for (size_t r = 0; r < sequences.size(); ++r) {
    int best = -1;
    if (!best_read_alignment(r, &best)) {
      out[r].status = force_alignment_ ? 2 : 0;
      continue;
    }
    const HaplotypeAlignment& ha = alignments_[best];
    const ReadAlignment& ra = ha.reads[r];
    if (ra.position >= ha.hap_to_ref.size()) continue;   // CHECK in the reference
    const int shift = ha.hap_to_ref[ra.position];
    Cigar ops;
    std::string error;
    if (!calculate_read_to_ref_alignment(r, ra, ha.cigar_ops, &ops, &error)) ops.clear();
    const int64_t offset = static_cast<int64_t>(ha.ref_pos) + ra.position + shift;
    if (!normalize_reads_ && !is_alignment_normalized(ops, static_cast<int>(offset), reads_[r])) ops.clear();
    if (!ops.empty()) {
      out[r].status = 1;
      out[r].position = static_cast<int64_t>(region_position_) + offset;
      out[r].cigar = std::move(ops);
    }
  }
  return out;
}

}  // namespace dv
