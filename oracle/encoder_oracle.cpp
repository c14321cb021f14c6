// encoder_oracle.cpp -- CPU ORACLE for the pileup-image encoder.
//
// TEST INFRASTRUCTURE ONLY (see dvo.h).  A from-scratch restatement of the
// reference algorithm of google/deepvariant v1.10.0; every function cites the
// reference file:line it follows.  Parity status: PINNED against the
// reference's known-answer vectors (pileup_image_native_test.cc:277-413,
// pileup_image_test.py:138-785, pileup_channel_lib_test.cc) and its golden
// TFRecords (tests/test_oracle_*.py, tests/golden/).
//
// Build: see oracle/Makefile (g++ -O2 -std=c++17, no -ffast-math: the
// float->int truncations below must be evaluated in IEEE fp32 exactly like
// the reference's static_cast<int>(254.0f * x)).

#include "absl_uniform_restated.h"
#include "dvo.h"
#include "packed_adapter.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <numeric>
#include <optional>
#include <random>
#include <set>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

namespace {

thread_local std::string g_error;

int fail(const std::string& msg) {
  g_error = msg;
  return -1;
}

// channels/channel.h:78-81
constexpr float kMaxPixelValueAsFloat = 254.0f;
constexpr float kMaxFragmentLength = 1000.0f;
// pileup_image_native.h (kChannelValue255 / kChannelValue200)
constexpr unsigned char kChannelValue255 = 255;
constexpr unsigned char kChannelValue200 = 200;

using Row = std::vector<std::vector<unsigned char>>;  // channel_data[C][W]

// The "ScaleColor" every channel re-declares, e.g.
// channels/base_quality_channel.cc:59-66.
inline std::uint8_t ScaleColor(int value, float max_val) {
  if (static_cast<float>(value) > max_val) {
    value = max_val;
  }
  return static_cast<int>(kMaxPixelValueAsFloat *
                          (static_cast<float>(value) / max_val));
}

// channels/is_homopolymer_channel.cc:101-113 (ScaleColorVector)
void ScaleColorVector(std::vector<std::uint8_t>& v, float max_val) {
  for (size_t i = 0; i < v.size(); i++) {
    int value = v[i];
    if (static_cast<float>(value) > max_val) {
      value = max_val;
    }
    v[i] = static_cast<int>(kMaxPixelValueAsFloat *
                            (static_cast<float>(value) / max_val));
  }
}

std::string ReadKey(const dvo_read& read) {
  // channels/read_supports_variant_channel.cc:78-79
  return std::string(read.fragment_name) + "/" +
         std::to_string(read.read_number);
}

// Per-read channel state: the reference allocates one Channel object per
// channel per read (pileup_channel_lib.cc:109-114) and several of them cache a
// value on first use; the caches below play that role.
struct ReadChannels {
  const dvo_options& opt;
  const dvo_call* call;
  const dvo_read* read;
  const char* const* alt_alleles;
  int n_alt_alleles;

  std::optional<unsigned char> supports_variant_color;
  std::optional<unsigned char> fuzzy_color;
  std::optional<unsigned char> insert_size_color;
  std::optional<unsigned char> haplotype_color;
  std::optional<unsigned char> allele_frequency_color;
  std::optional<unsigned char> mapping_percent_color;
  std::optional<unsigned char> avg_bq_color;
  std::optional<unsigned char> identity_color;
  std::optional<unsigned char> gc_identity_color;
  std::optional<unsigned char> gc_content_color;
  std::optional<std::vector<std::uint8_t>> is_homopolymer;
  std::optional<std::vector<std::uint8_t>> homopolymer_weighted;
  std::optional<std::vector<std::uint8_t>> methylation;
  std::optional<std::vector<std::uint8_t>> m6a;
  std::optional<std::vector<std::uint8_t>> hmer_insertion_quality;
  std::optional<std::vector<std::uint8_t>> hmer_deletion_quality;
  std::optional<std::vector<std::uint8_t>> t0_quality;
  bool error = false;

  // channels/read_base_channel.cc:56-73
  int BaseColor(char base) const {
    switch (base) {
      case 'A':
        return opt.base_color_offset_a_and_g + opt.base_color_stride * 3;
      case 'G':
        return opt.base_color_offset_a_and_g + opt.base_color_stride * 2;
      case 'T':
        return opt.base_color_offset_t_and_c + opt.base_color_stride * 1;
      case 'C':
        return opt.base_color_offset_t_and_c + opt.base_color_stride * 0;
      default:
        return 0;
    }
  }

  // channels/base_differs_from_ref_channel.cc:59-66
  int MatchesRefColor(bool matches) const {
    float alpha = matches ? opt.reference_matching_read_alpha
                          : opt.reference_mismatching_read_alpha;
    return static_cast<int>(kMaxPixelValueAsFloat * alpha);
  }

  bool InAltAlleles(const std::string& alt) const {
    for (int i = 0; i < n_alt_alleles; ++i) {
      if (alt == alt_alleles[i]) return true;
    }
    return false;
  }

  int FindSupport(const std::string& allele) const {
    for (int s = 0; s < call->n_support; ++s) {
      if (allele == call->support_alleles[s]) return s;
    }
    return -1;
  }

  // channels/read_supports_variant_channel.cc:75-104
  int ReadSupportsAlt() const {
    const std::string key = ReadKey(*read);
    for (int a = 0; a < call->n_alts; ++a) {
      const std::string alt_allele = call->alts[a];
      const int s = FindSupport(alt_allele);
      if (s >= 0) {
        for (int n = call->support_offsets[s]; n < call->support_offsets[s + 1];
             ++n) {
          const bool alt_in_alt_alleles = InAltAlleles(alt_allele);
          if (key == call->support_names[n] && alt_in_alt_alleles) {
            return 1;
          } else if (key == call->support_names[n] && !alt_in_alt_alleles) {
            return 2;
          }
        }
      }
    }
    return 0;
  }

  // channels/read_supports_variant_channel.cc:105-116
  int SupportsAltColor(int read_supports_alt) const {
    float alpha;
    if (read_supports_alt == 0) {
      alpha = opt.allele_unsupporting_read_alpha;
    } else if (read_supports_alt == 1) {
      alpha = opt.allele_supporting_read_alpha;
    } else {
      alpha = opt.other_allele_supporting_read_alpha;
    }
    return static_cast<int>(kMaxPixelValueAsFloat * alpha);
  }

  // ---- channels/read_supports_variant_fuzzy_channel.cc ---------------------
  // CalculateAlelePhases (:78-99): ALT_PS value i+1 is alt allele i's phase.
  std::vector<int> FuzzyAllelePhases(int num_alt_alleles) const {
    std::vector<int> phases(num_alt_alleles, 0);
    if (call->alt_ps_present) {
      for (int i = 0; i < num_alt_alleles; ++i) {
        phases[i] = call->n_alt_ps > i + 1 ? call->alt_ps[i + 1] : 0;
      }
    }
    return phases;
  }

  // CalculateReadSupport (:119-186) over one allele's supporting read names.
  int FuzzyReadSupport(const std::string& alt_allele, const char* const* names, int n_names,
                       const std::string& key, const std::vector<int>& phases) const {
    for (int n = 0; n < n_names; ++n) {
      const std::string read_name = names[n];
      const bool alt_in_alt_alleles = InAltAlleles(alt_allele);
      if (read_name == key && alt_in_alt_alleles) {
        return 1;
      } else if (read_name == key && !alt_in_alt_alleles) {
        for (int ia = 0; ia < n_alt_alleles; ++ia) {
          int global = 0;
          for (int a = 0; a < call->n_alts; ++a) {
            if (std::string(call->alts[a]) == alt_alleles[ia]) break;
            ++global;
          }
          if (global >= static_cast<int>(phases.size())) {  // CHECK_LT (:147)
            const_cast<ReadChannels*>(this)->error = true;
            return 0;
          }
          int hp_value = 0;  // read.info["HP"].values(0).int_value(), 0 if absent (:149-155)
          if (read->hp_present && read->hp_n_values > 0) {
            hp_value = read->hp_is_int ? read->hp_value : 0;
          }
          if (phases[global] == 0 || hp_value == 0 ||
              (phases[global] == hp_value && hp_value != 0)) {
            const int d = std::abs(static_cast<int>(strlen(alt_alleles[ia])) -
                                   static_cast<int>(alt_allele.size()));
            if (d == 1) return 10;  // kFuzzySupportValueOneBaseDifference
            if (d == 2) return 9;   // kFuzzySupportValueTwoBasesDifference
          }
        }
        return 2;
      }
    }
    return 0;
  }

  // ReadSupportsAlt (:205-288): called alts, then rejected alts, then the reference allele.
  int FuzzyReadSupportsAlt() const {
    const std::string key = ReadKey(*read);
    const std::vector<int> phases = FuzzyAllelePhases(call->n_alts);
    for (int a = 0; a < call->n_alts; ++a) {
      const std::string alt_allele = call->alts[a];
      const int s = FindSupport(alt_allele);
      if (s >= 0) {
        const int rs = FuzzyReadSupport(
            alt_allele, call->support_names + call->support_offsets[s],
            call->support_offsets[s + 1] - call->support_offsets[s], key, phases);
        if (rs == 1 || rs == 10 || rs == 9) return rs;
      }
    }
    for (int a = 0; a < call->n_rejected_alts; ++a) {
      const std::string alt_allele = call->rejected_alts[a];
      for (int s = 0; s < call->n_rejected_support; ++s) {
        if (alt_allele != call->rejected_support_alleles[s]) continue;
        const int rs = FuzzyReadSupport(
            alt_allele, call->rejected_support_names + call->rejected_support_offsets[s],
            call->rejected_support_offsets[s + 1] - call->rejected_support_offsets[s], key,
            phases);
        if (rs != 0) return rs;
        break;
      }
    }
    if (call->n_ref_support > 0 && call->ref_support_names != nullptr) {
      const std::string ref = call->reference_bases ? call->reference_bases : "";
      const int rs = FuzzyReadSupport(ref, call->ref_support_names, call->n_ref_support, key,
                                      phases);
      if (rs == 10 || rs == 9) return rs;
    }
    return 0;
  }

  // SupportsAltColor (:290-312)
  int FuzzySupportsAltColor(int read_supports_alt) const {
    float alpha;
    if (read_supports_alt == 0) {
      alpha = opt.allele_unsupporting_read_alpha;
    } else if (read_supports_alt == 1) {
      alpha = opt.allele_supporting_read_alpha;
    } else if (read_supports_alt == 10) {
      alpha = 0.90;  // kReadSupportAltWithinOneBase (a float in the reference)
    } else if (read_supports_alt == 9) {
      alpha = 0.80;  // kReadSupportAltWithinTwoBases
    } else if (read_supports_alt == 8) {
      alpha = 0.70;
    } else {
      alpha = opt.other_allele_supporting_read_alpha;
    }
    return static_cast<int>(kMaxPixelValueAsFloat * alpha);
  }

  // channels/insert_size_channel.cc:79-90
  int NormalizeFragmentLength() const {
    int fragment_length = std::abs(read->fragment_length);
    if (static_cast<float>(fragment_length) > kMaxFragmentLength) {
      fragment_length = static_cast<int>(kMaxFragmentLength);
    }
    return static_cast<int>(
        kMaxPixelValueAsFloat *
        (static_cast<float>(fragment_length) / kMaxFragmentLength));
  }

  // channels/haplotype_tag_channel.cc:76-100
  int HPValueForHPChannel() const {
    if (!read->hp_present) return 0;
    if (read->hp_n_values == 0) return 0;
    if (read->hp_n_values > 1) return 0;
    int hp_value = read->hp_is_int ? read->hp_value : 0;
    if (opt.hp_tag_for_assembly_polishing == 2) {
      if (hp_value == 1) return 2;
      if (hp_value == 2) return 1;
    }
    return hp_value;
  }

  // channels/allele_frequency_channel.cc:74-86
  unsigned char AlleleFrequencyColor(float allele_frequency) const {
    if (allele_frequency <= opt.min_non_zero_allele_frequency) {
      return 0;
    } else {
      float log10_af = log10(allele_frequency);
      float log10_min = log10(opt.min_non_zero_allele_frequency);
      return ((log10_min - log10_af) / log10_min) *
             static_cast<int>(kMaxPixelValueAsFloat);
    }
  }

  // channels/allele_frequency_channel.cc:89-119
  float ReadAlleleFrequency() const {
    const std::string key = ReadKey(*read);
    for (int a = 0; a < call->n_alts; ++a) {
      const std::string alt_allele = call->alts[a];
      const int s = FindSupport(alt_allele);
      if (s >= 0) {
        for (int n = call->support_offsets[s]; n < call->support_offsets[s + 1];
             ++n) {
          if (key == call->support_names[n] && InAltAlleles(alt_allele)) {
            for (int f = 0; f < call->n_af; ++f) {
              if (alt_allele == call->af_alleles[f]) return call->af_values[f];
            }
            return 0;
          }
        }
      }
    }
    return 0;
  }

  // channels/read_mapping_percent_channel.cc:67-88, identity_channel.cc:66-93
  // (identical arithmetic: M and = count as matches).
  int MatchPercent() const {
    int match_len = 0;
    for (int i = 0; i < read->n_cigar; ++i) {
      int op = read->cigar_ops[i];
      if (op == DVO_CIGAR_SEQUENCE_MATCH || op == DVO_CIGAR_ALIGNMENT_MATCH) {
        match_len += static_cast<int>(read->cigar_lens[i]);
      }
    }
    float mapping_percent = (static_cast<float>(match_len) /
                             static_cast<float>(read->seq_len)) *
                            100;
    return static_cast<int>(mapping_percent);
  }

  // channels/gap_compressed_identity_channel.cc:68-104
  int GapCompressedIdentity() const {
    int match_len = 0;
    int gap_compressed_len = 0;
    for (int i = 0; i < read->n_cigar; ++i) {
      int op = read->cigar_ops[i];
      int op_len = static_cast<int>(read->cigar_lens[i]);
      switch (op) {
        case DVO_CIGAR_SEQUENCE_MATCH:
        case DVO_CIGAR_ALIGNMENT_MATCH:
          match_len += op_len;
          gap_compressed_len += op_len;
          break;
        case DVO_CIGAR_SEQUENCE_MISMATCH:
          gap_compressed_len += op_len;
          break;
        case DVO_CIGAR_INSERT:
          gap_compressed_len += 1;
          break;
        case DVO_CIGAR_DELETE:
          gap_compressed_len += 1;
          break;
        default:
          break;
      }
    }
    float v = static_cast<float>(match_len) /
              static_cast<float>(gap_compressed_len) * 100;
    return static_cast<int>(v);
  }

  // channels/avg_base_quality_channel.cc:79-95
  int AvgBaseQuality() {
    int base_qual_sum = 0;
    for (int i = 0; i < read->qual_len; ++i) {
      int q = read->qual[i];
      base_qual_sum += q;
      if (q < 0 || q > 93) {
        error = true;  // reference: LOG(FATAL)
      }
    }
    float avg = static_cast<float>(base_qual_sum) /
                static_cast<float>(read->qual_len);
    return static_cast<int>(avg);
  }

  static int GcContent(const char* seq, int n) {
    // channels/gc_content_channel.cc:84-94
    int gc_count = 0;
    for (int i = 0; i < n; ++i) {
      if (seq[i] == 'G' || seq[i] == 'C') gc_count += 1;
    }
    return static_cast<int>(
        (static_cast<float>(gc_count) / static_cast<float>(n)) * 100);
  }

  static std::vector<std::uint8_t> IsHomopolymer(const char* seq, int n) {
    // channels/is_homopolymer_channel.cc:83-98
    std::vector<std::uint8_t> h(n, 0);
    for (int i = 2; i < n; i++) {
      if (seq[i] == seq[i - 1] && seq[i - 1] == seq[i - 2]) {
        h[i] = 1;
        h[i - 1] = 1;
        h[i - 2] = 1;
      }
    }
    return h;
  }

  static std::vector<std::uint8_t> HomopolymerWeighted(const char* seq, int n) {
    // channels/homopolymer_weighted_channel.cc:85-108
    std::vector<std::uint8_t> h(n, 0);
    int current_weight = 1;
    for (int i = 1; i < n; i++) {
      if (seq[i] == seq[i - 1]) {
        current_weight += 1;
      } else {
        for (int cw = current_weight; cw >= 1; cw--) {
          h[i - cw] = current_weight;
        }
        current_weight = 1;
      }
    }
    for (int cw = current_weight; cw >= 1; cw--) {
      if (n - cw >= 0) h[n - cw] = current_weight;
    }
    return h;
  }

  // One FillReadBase per channel class (channels/*.cc); returns false for an
  // unsupported channel.
  bool FillReadBase(int ch, std::vector<unsigned char>& data, int col,
                    char read_base, char ref_base, int base_quality,
                    int read_index) {
    switch (ch) {
      case DVO_CH_READ_BASE:  // read_base_channel.cc:43-49
        data[col] = BaseColor(read_base);
        return true;
      case DVO_CH_BASE_QUALITY:  // base_quality_channel.cc:44-50
        data[col] = ScaleColor(base_quality, opt.base_quality_cap);
        return true;
      case DVO_CH_MAPPING_QUALITY:  // mapping_quality_channel.cc:43-51
        data[col] = ScaleColor(read->mapping_quality, opt.mapping_quality_cap);
        return true;
      case DVO_CH_STRAND: {  // strand_channel.cc:44-61
        bool fwd = !read->reverse_strand;
        data[col] = static_cast<std::uint8_t>(fwd ? opt.positive_strand_color
                                                  : opt.negative_strand_color);
        return true;
      }
      case DVO_CH_READ_SUPPORTS_VARIANT:  // read_supports_variant_channel.cc:54-66
        if (!supports_variant_color.has_value()) {
          supports_variant_color =
              static_cast<unsigned char>(SupportsAltColor(ReadSupportsAlt()));
        }
        data[col] = supports_variant_color.value();
        return true;
      case DVO_CH_READ_SUPPORTS_VARIANT_FUZZY:  // read_supports_variant_fuzzy_channel.cc:101-113
        if (!fuzzy_color.has_value()) {
          fuzzy_color = static_cast<unsigned char>(FuzzySupportsAltColor(FuzzyReadSupportsAlt()));
        }
        data[col] = fuzzy_color.value();
        return true;
      case DVO_CH_BASE_DIFFERS_FROM_REF:  // base_differs_from_ref_channel.cc:43-51
        data[col] = MatchesRefColor(read_base == ref_base);
        return true;
      case DVO_CH_HAPLOTYPE_TAG:  // haplotype_tag_channel.cc:54-67
        if (!haplotype_color.has_value()) {
          haplotype_color = ScaleColor(HPValueForHPChannel(), 2);
        }
        data[col] = haplotype_color.value();
        return true;
      case DVO_CH_ALLELE_FREQUENCY:  // allele_frequency_channel.cc:53-64
        if (!allele_frequency_color.has_value()) {
          allele_frequency_color = AlleleFrequencyColor(ReadAlleleFrequency());
        }
        data[col] = allele_frequency_color.value();
        return true;
      case DVO_CH_READ_MAPPING_PERCENT:  // read_mapping_percent_channel.cc:49-59
        if (!mapping_percent_color.has_value()) {
          mapping_percent_color = ScaleColor(MatchPercent(), 100);
        }
        data[col] = mapping_percent_color.value();
        return true;
      case DVO_CH_AVG_BASE_QUALITY:  // avg_base_quality_channel.cc:49-59
        if (!avg_bq_color.has_value()) {
          avg_bq_color = ScaleColor(AvgBaseQuality(), 93);
        }
        data[col] = avg_bq_color.value();
        return true;
      case DVO_CH_IDENTITY:  // identity_channel.cc:49-58
        if (!identity_color.has_value()) {
          identity_color = ScaleColor(MatchPercent(), 100);
        }
        data[col] = identity_color.value();
        return true;
      case DVO_CH_GAP_COMPRESSED_IDENTITY:  // gap_compressed_identity_channel.cc:50-60
        if (!gc_identity_color.has_value()) {
          gc_identity_color = ScaleColor(GapCompressedIdentity(), 100);
        }
        data[col] = gc_identity_color.value();
        return true;
      case DVO_CH_GC_CONTENT:  // gc_content_channel.cc:51-61
        if (!gc_content_color.has_value()) {
          gc_content_color =
              ScaleColor(GcContent(read->seq, read->seq_len), 100);
        }
        data[col] = gc_content_color.value();
        return true;
      case DVO_CH_IS_HOMOPOLYMER:  // is_homopolymer_channel.cc:52-64
        if (!is_homopolymer.has_value()) {
          is_homopolymer = IsHomopolymer(read->seq, read->seq_len);
          ScaleColorVector(*is_homopolymer, 1);
        }
        data[col] = is_homopolymer->at(read_index);
        return true;
      case DVO_CH_HOMOPOLYMER_WEIGHTED:  // homopolymer_weighted_channel.cc:52-66
        if (!homopolymer_weighted.has_value()) {
          homopolymer_weighted = HomopolymerWeighted(read->seq, read->seq_len);
          ScaleColorVector(*homopolymer_weighted, 30);
        }
        data[col] = homopolymer_weighted->at(read_index);
        return true;
      case DVO_CH_BLANK:          // blank_channel.cc:43-50
      case DVO_CH_MEAN_COVERAGE:  // pileup_channel_lib.cc:379-383 (BlankChannel)
        data[col] = 0;
        return true;
      case DVO_CH_INSERT_SIZE:  // insert_size_channel.cc:55-66
        if (!insert_size_color.has_value()) {
          insert_size_color =
              static_cast<std::uint8_t>(NormalizeFragmentLength());
        }
        data[col] = insert_size_color.value();
        return true;
      case DVO_CH_BASE_METHYLATION:  // base_methylation_channel.cc:52-66
        if (!methylation.has_value()) {
          methylation = std::vector<std::uint8_t>();
          if (read->mod_5mc != nullptr) {
            methylation->assign(read->mod_5mc,
                                read->mod_5mc + read->mod_5mc_len);
          }
          ScaleColorVector(*methylation, 255);
        }
        if (!methylation->empty()) data[col] = methylation->at(read_index);
        return true;
      case DVO_CH_BASE_6MA:  // base_6ma_channel.cc:52-66
        if (!m6a.has_value()) {
          m6a = std::vector<std::uint8_t>();
          if (read->mod_6ma != nullptr) {
            m6a->assign(read->mod_6ma, read->mod_6ma + read->mod_6ma_len);
          }
          ScaleColorVector(*m6a, 255);
        }
        if (!m6a->empty()) data[col] = m6a->at(read_index);
        return true;
      case DVO_CH_HOMOPOLYMER_INSERTION_QUALITY:  // homopolymer_insertion_quality_channel.cc:46-61
        if (!hmer_insertion_quality.has_value()) hmer_insertion_quality = HomoPolymerInDelQuality(false);
        data[col] = read_index >= 0 && read_index < static_cast<int>(hmer_insertion_quality->size())
                        ? (*hmer_insertion_quality)[read_index] : 0;
        return true;
      case DVO_CH_HOMOPOLYMER_DELETION_QUALITY:  // homopolymer_deletion_quality_channel.cc:46-61
        if (!hmer_deletion_quality.has_value()) hmer_deletion_quality = HomoPolymerInDelQuality(true);
        data[col] = read_index >= 0 && read_index < static_cast<int>(hmer_deletion_quality->size())
                        ? (*hmer_deletion_quality)[read_index] : 0;
        return true;
      case DVO_CH_INTER_HOMOPOLYMER_INSERTION_QUALITY:  // inter_homopolymer_insertion_quality_channel.cc:53-68
        if (!t0_quality.has_value()) t0_quality = T0QualityValues();
        data[col] = read_index >= 0 && read_index < static_cast<int>(t0_quality->size())
                        ? (*t0_quality)[read_index] : 0;
        return true;
      case DVO_CH_SUPPLEMENTARY_ALIGNMENT: {  // supplementary_alignment_channel.cc:49-59
        float alpha = read->supplementary ? opt.allele_supporting_read_alpha
                                          : opt.allele_unsupporting_read_alpha;
        data[col] = static_cast<unsigned char>(kMaxPixelValueAsFloat * alpha);
        return true;
      }
      case DVO_CH_ALLELE_SAMPLE_PROBABILITY: {
        // allele_sample_probability_channel.cc:43-75.  NOTE: the reference
        // iterates a proto map (hash order, unspecified); the oracle walks the
        // alleles in KEY order -- what the same code does over an ordered map
        // (the reference build of oracle/ref_build).  The two only have to
        // agree where the early `break` cannot change `total_reads` (a single
        // allele, or a read in no allele).
        int total_reads = 0;
        int total_reads_supporting_allele = 0;
        const std::string read_key = ReadKey(*read);
        bool found = false;
        std::vector<int> order(static_cast<size_t>(call->n_support));
        std::iota(order.begin(), order.end(), 0);
        std::sort(order.begin(), order.end(), [&](int a, int b) {
          return std::string(call->support_alleles[a]) < call->support_alleles[b];
        });
        for (int oi = 0; oi < call->n_support && !found; ++oi) {
          const int s = order[static_cast<size_t>(oi)];
          const int n0 = call->support_offsets[s];
          const int n1 = call->support_offsets[s + 1];
          total_reads += n1 - n0;
          for (int n = n0; n < n1; ++n) {
            if (read_key == call->support_names[n]) {
              total_reads_supporting_allele = n1 - n0;
              found = true;
              break;
            }
          }
        }
        if (!found) total_reads_supporting_allele = call->n_ref_support;
        total_reads += call->n_ref_support;
        data[col] = SampleProbabilityColor(total_reads_supporting_allele,
                                           total_reads);
        return true;
      }
      default:
        return false;
    }
  }

  // channels/channel_utils.cc:41-44 with channel_utils.h:44-49's OWN constants: the maximum pixel is 255.0 there
  // (254.0 everywhere else in the encoder), the maximum quality the float 93.0
  static std::uint8_t BaseQualityColor(int base_qual) {
    return static_cast<std::uint8_t>(255.0f * base_qual / 93.0f);
  }

  // HomopolymerInDelQualityChannel::HomoPolymerInDelQuality with GetTPValues and HomoPolymerWeighted,
  // homopolymer_indel_quality_channel.cc:68-183 (Ultima: QUAL[i] is the phred score of the homopolymer being
  // tp[i] longer / shorter than called; the errors of one direction are summed over the homopolymer).
  std::vector<std::uint8_t> HomoPolymerInDelQuality(bool is_deletion) const {
    const int kMaxQScore = 93;  // homopolymer_indel_quality_channel.h:65
    const size_t n = static_cast<size_t>(read->seq_len);
    std::vector<std::uint8_t> out(n, BaseQualityColor(kMaxQScore));
    std::vector<std::uint8_t> hmer(n, 1);  // :86-120, run length of the homopolymer a base belongs to, capped at 255
    for (size_t i = 0; i < n;) {
      size_t j = i + 1;
      while (j < n && read->seq[j] == read->seq[i]) ++j;
      for (size_t k = i; k < j; ++k) hmer[k] = static_cast<std::uint8_t>(std::min<size_t>(j - i, 255));
      i = j;
    }
    std::vector<int8_t> tps(n, 0);  // :68-84
    if (read->tp != nullptr) {
      for (size_t i = 0; i < static_cast<size_t>(read->tp_len) && i < n; ++i) tps[i] = read->tp[i];
    }
    if (tps.empty()) return out;  // :137-139 (the sizes are equal by construction)
    size_t i = 0;
    while (i < n) {
      const int len = hmer[i];
      float err = 0;
      for (int j = 0; j < len; ++j) {
        if (tps[i + j] == 0) continue;
        if ((tps[i + j] < 0) == is_deletion) {
          const std::uint8_t q = read->qual[i + j];
          const float e = std::pow(10, (q / -10.0));
          err += e;
        }
      }
      int hq = err == 0 ? kMaxQScore : static_cast<int>(-10 * std::log10(err));
      if (hq > kMaxQScore) hq = kMaxQScore;
      for (int j = 0; j < len; ++j) out[i + j] = BaseQualityColor(hq);
      i += static_cast<size_t>(len);
    }
    return out;
  }

  // InterHomopolymerInsertionQualityChannel::GetT0QualityValues / GetT0Values,
  // inter_homopolymer_insertion_quality_channel.cc:76-125.
  std::vector<std::uint8_t> T0QualityValues() const {
    std::vector<std::uint8_t> out(static_cast<size_t>(read->seq_len), 0);
    if (read->t0 != nullptr) {
      for (size_t i = 0; i < static_cast<size_t>(read->t0_len) && i < out.size(); ++i) {
        out[i] = static_cast<std::uint8_t>(read->t0[i] - 33);
      }
    }
    for (std::uint8_t& v : out) v = BaseQualityColor(v);
    return out;
  }

  // allele_sample_probability_channel.cc:84-98
  static std::uint8_t SampleProbabilityColor(int value, float max_val) {
    if (max_val == 0) return 0;
    float value_as_float = static_cast<float>(value);
    value_as_float = std::clamp<float>(value_as_float, 0.0f, max_val);
    double probability = value_as_float / max_val;
    double scaled_probability = std::sqrt(probability);
    return static_cast<int>(kMaxPixelValueAsFloat * scaled_probability);
  }
};

// FillRefBase of every channel class, channels/*.cc.
bool FillRefBase(const dvo_options& opt, int ch,
                 std::vector<unsigned char>& ref_data, int col, char ref_base,
                 const std::string& ref_bases,
                 std::optional<unsigned char>& gc_cache,
                 std::optional<std::vector<std::uint8_t>>& vec_cache) {
  ReadChannels rc{opt, nullptr, nullptr, nullptr, 0};
  switch (ch) {
    case DVO_CH_READ_BASE:  // read_base_channel.cc:51-54
      ref_data[col] = rc.BaseColor(ref_base);
      return true;
    case DVO_CH_BASE_QUALITY:     // base_quality_channel.cc:52-57
    case DVO_CH_MAPPING_QUALITY:  // mapping_quality_channel.cc:53-58 (uses base_quality_cap)
      ref_data[col] = ScaleColor(opt.reference_base_quality,
                                 opt.base_quality_cap);
      return true;
    case DVO_CH_STRAND:  // strand_channel.cc:53-56
      ref_data[col] = static_cast<std::uint8_t>(opt.positive_strand_color);
      return true;
    case DVO_CH_READ_SUPPORTS_VARIANT:  // read_supports_variant_channel.cc:68-72
      ref_data[col] = rc.SupportsAltColor(0);
      return true;
    case DVO_CH_READ_SUPPORTS_VARIANT_FUZZY:  // read_supports_variant_fuzzy_channel.cc:115-119
      ref_data[col] = rc.FuzzySupportsAltColor(0);
      return true;
    case DVO_CH_BASE_DIFFERS_FROM_REF:  // base_differs_from_ref_channel.cc:53-57
      ref_data[col] = rc.MatchesRefColor(true);
      return true;
    case DVO_CH_HAPLOTYPE_TAG:  // haplotype_tag_channel.cc:69-73
      ref_data[col] = ScaleColor(0, 2);
      return true;
    case DVO_CH_ALLELE_FREQUENCY:  // allele_frequency_channel.cc:66-70
      ref_data[col] = rc.AlleleFrequencyColor(0);
      return true;
    case DVO_CH_READ_MAPPING_PERCENT:      // read_mapping_percent_channel.cc:61-65
    case DVO_CH_AVG_BASE_QUALITY:          // avg_base_quality_channel.cc:61-66
    case DVO_CH_IDENTITY:                  // identity_channel.cc:60-63
    case DVO_CH_GAP_COMPRESSED_IDENTITY:   // gap_compressed_identity_channel.cc:62-66
    case DVO_CH_INSERT_SIZE:               // insert_size_channel.cc:68-72
      ref_data[col] = static_cast<std::uint8_t>(kMaxPixelValueAsFloat);
      return true;
    case DVO_CH_GC_CONTENT:  // gc_content_channel.cc:63-73
      if (!gc_cache.has_value()) {
        gc_cache = ScaleColor(
            ReadChannels::GcContent(ref_bases.data(), ref_bases.size()), 100);
      }
      ref_data[col] = gc_cache.value();
      return true;
    case DVO_CH_IS_HOMOPOLYMER:  // is_homopolymer_channel.cc:66-78
      if (!vec_cache.has_value()) {
        vec_cache =
            ReadChannels::IsHomopolymer(ref_bases.data(), ref_bases.size());
        ScaleColorVector(*vec_cache, 1);
      }
      ref_data[col] = vec_cache->at(col);
      return true;
    case DVO_CH_HOMOPOLYMER_WEIGHTED:  // homopolymer_weighted_channel.cc:68-82
      if (!vec_cache.has_value()) {
        vec_cache = ReadChannels::HomopolymerWeighted(ref_bases.data(),
                                                      ref_bases.size());
        ScaleColorVector(*vec_cache, 30);
      }
      ref_data[col] = vec_cache->at(col);
      return true;
    case DVO_CH_BLANK:
    case DVO_CH_MEAN_COVERAGE:
    case DVO_CH_ALLELE_SAMPLE_PROBABILITY:  // allele_sample_probability_channel.cc:77-81
    case DVO_CH_BASE_METHYLATION:  // base_methylation_channel.cc:68-72
    case DVO_CH_BASE_6MA:          // base_6ma_channel.cc:68-72
    // the three Ultima channels push_back(0) behind the `width` zeros the row starts with
    // (homopolymer_insertion_quality_channel.cc:63-67 and its two siblings): column `col` stays 0
    case DVO_CH_HOMOPOLYMER_INSERTION_QUALITY:
    case DVO_CH_HOMOPOLYMER_DELETION_QUALITY:
    case DVO_CH_INTER_HOMOPOLYMER_INSERTION_QUALITY:
      ref_data[col] = 0;
      return true;
    case DVO_CH_SUPPLEMENTARY_ALIGNMENT:  // supplementary_alignment_channel.cc:61-65
      // The reference assigns the alpha float itself (0.6 -> 0).
      ref_data[col] = opt.allele_unsupporting_read_alpha;
      return true;
    default:
      return false;
  }
}

bool IsBlanked(int ch, const int32_t* blank, int n_blank) {
  for (int i = 0; i < n_blank; ++i) {
    if (blank[i] == ch) return true;
  }
  return false;
}

// Channels::CalculateChannels + CalculateBaseLevelData,
// pileup_channel_lib.cc:91-261.  Returns 1 ok, 0 rejected, -1 error.
int CalculateChannels(const dvo_options& opt, Row& data, const dvo_read& read,
                      const std::string& ref_bases, const dvo_call& call,
                      const char* const* alt_alleles, int n_alt_alleles,
                      int image_start_pos, const int32_t* blank, int n_blank) {
  ReadChannels rc{opt, &call, &read, alt_alleles, n_alt_alleles};
  bool unsupported = false;

  // The per-base action, pileup_channel_lib.cc:126-165.
  auto action = [&](int ref_i, int read_i, int cigar_op) -> bool {
    char read_base = 0;
    if (cigar_op == DVO_CIGAR_INSERT) {
      read_base = static_cast<char>(opt.indel_anchoring_base_char);
    } else if (cigar_op == DVO_CIGAR_DELETE) {
      ref_i -= 1;  // anchor base on the reference
      read_base = static_cast<char>(opt.indel_anchoring_base_char);
    } else if (cigar_op == DVO_CIGAR_ALIGNMENT_MATCH ||
               cigar_op == DVO_CIGAR_SEQUENCE_MATCH ||
               cigar_op == DVO_CIGAR_SEQUENCE_MISMATCH) {
      read_base = read.seq[read_i];
    }
    size_t col = ref_i - image_start_pos;
    if (read_base && col < ref_bases.size()) {
      uint8_t base_quality = read.qual[read_i];
      if (ref_i == call.variant_start &&
          base_quality < opt.min_base_quality) {
        return false;
      }
      char ref_base = ref_bases[col];
      for (int c = 0; c < opt.n_channels; ++c) {
        const int ch = opt.channels[c];
        if (!IsBlanked(ch, blank, n_blank)) {
          if (!rc.FillReadBase(ch, data[c], col, read_base, ref_base,
                               base_quality, read_i)) {
            unsupported = true;
          }
        }
      }
    }
    return true;
  };

  // CalculateBaseLevelData, pileup_channel_lib.cc:171-261.
  int ref_i = static_cast<int>(read.position);
  int read_i = 0;
  bool ok = true;
  for (int k = 0; k < read.n_cigar; ++k) {
    const int op = read.cigar_ops[k];
    const int op_len = static_cast<int>(read.cigar_lens[k]);
    switch (op) {
      case DVO_CIGAR_ALIGNMENT_MATCH:
      case DVO_CIGAR_SEQUENCE_MATCH:
      case DVO_CIGAR_SEQUENCE_MISMATCH:
        for (int i = 0; i < op_len; i++) {
          ok = ok && action(ref_i, read_i, op);
          ref_i++;
          read_i++;
        }
        break;
      case DVO_CIGAR_INSERT:
      case DVO_CIGAR_CLIP_SOFT:
        if (ref_i > 0) {
          ok = action(ref_i - 1, read_i, op);
        }
        read_i += op_len;
        break;
      case DVO_CIGAR_DELETE:
      case DVO_CIGAR_SKIP:
        if (read_i > 0) {
          ok = action(ref_i, read_i - 1, op);
        }
        ref_i += op_len;
        break;
      case DVO_CIGAR_CLIP_HARD:
      case DVO_CIGAR_PAD:
        break;
      default:
        g_error = "Unrecognized CIGAR op";  // reference: LOG(FATAL)
        return -1;
    }
    if (!ok) {
      return 0;
    }
  }
  if (unsupported) {
    g_error = "channel not implemented by the oracle";
    return -1;
  }
  if (rc.error) {
    g_error = "base quality outside of bounds (0,93)";
    return -1;
  }
  return 1;
}

// PileupImageEncoderNative::EncodeRead, pileup_image_native.cc:477-510.
int EncodeRead(const dvo_options& opt, const dvo_call& call,
               const std::string& ref_bases, const dvo_read& read,
               int image_start_pos, const char* const* alt_alleles,
               int n_alt_alleles, const int32_t* blank, int n_blank,
               Row* out) {
  if (read.mapping_quality < opt.min_mapping_quality) {
    return 0;
  }
  Row row(opt.n_channels, std::vector<unsigned char>(ref_bases.size(), 0));
  int rc = CalculateChannels(opt, row, read, ref_bases, call, alt_alleles,
                             n_alt_alleles, image_start_pos, blank, n_blank);
  if (rc != 1) return rc;
  *out = std::move(row);
  return 1;
}

// EncodeReference + Channels::CalculateRefRows,
// pileup_image_native.cc:512-527, pileup_channel_lib.cc:263-293.
int EncodeReference(const dvo_options& opt, const std::string& ref_bases,
                    Row* out) {
  Row row(opt.n_channels, std::vector<unsigned char>(ref_bases.size(), 0));
  for (int c = 0; c < opt.n_channels; ++c) {
    std::optional<unsigned char> gc_cache;
    std::optional<std::vector<std::uint8_t>> vec_cache;
    for (size_t i = 0; i < ref_bases.size(); ++i) {
      if (!FillRefBase(opt, opt.channels[c], row[c], i, ref_bases[i], ref_bases,
                       gc_cache, vec_cache)) {
        return fail("channel not implemented by the oracle");
      }
    }
  }
  *out = std::move(row);
  return 0;
}

// PileupImageEncoderNative::GetHapIndex, pileup_image_native.cc:449-475.
int GetHapIndex(const dvo_options& opt, const dvo_read& read) {
  if (!opt.sort_by_haplotypes || !read.hp_present) return 0;
  if (read.hp_n_values == 0) return 0;
  if (!read.hp_is_int) return 0;
  int hp_value = read.hp_value;
  if (opt.hp_tag_for_assembly_polishing > 0 &&
      hp_value == opt.hp_tag_for_assembly_polishing) {
    return -1;
  } else if (hp_value < 0) {
    return 0;
  }
  return hp_value;
}

// DownsampleReadIndices, pileup_image_native.cc:153-165.  The generator is
// passed by value in the reference, so every call restarts the stream.
std::vector<int> DownsampleReadIndices(int n, int max_reads, uint32_t seed) {
  std::vector<int> idx(n);
  std::iota(idx.begin(), idx.end(), 0);
  if (n > max_reads) {
    std::mt19937_64 gen(seed);
    std::shuffle(idx.begin(), idx.end(), gen);
  }
  return idx;
}

// GetReadIndicesAllelePartition + DownsampleReadIndicesWithMinsPerAllele (pileup_image_native.cc:242-294) over
// sampling::SampleWithPartitionMins / ReservoirSample (deepvariant/sampling_util.h:57-155): at least `min_per_allele`
// reads of every allele's supporters (and of the reads that support no allele) survive, the rest of the image is filled
// from what is left.  -> the kept read indices in ascending order (the reference returns a btree_set), or false where
// the reference returns an error (the thresholds alone exceed the image) and falls back to the uniform shuffle.
// A read listed under two alleles belongs to the one the proto map yields first: hash order in the reference, KEY order
// here and in the reference build (oracle/ref_build).  The draws follow oracle/absl_uniform_restated.h.
bool DownsampleWithMinsPerAllele(const dvo_call& call, const dvo_read* reads, int n_reads, int max_reads,
                                 int min_per_allele, uint32_t seed, std::vector<int>* out) {
  std::map<std::string, int> name_to_index;
  for (int i = 0; i < n_reads; ++i) name_to_index[ReadKey(reads[i])] = i;
  std::vector<int> order(static_cast<size_t>(call.n_support));
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int a, int b) {
    return std::string(call.support_alleles[a]) < call.support_alleles[b];
  });
  std::set<std::set<int>> partition;
  for (int s : order) {
    std::set<int> idx;
    for (int n = call.support_offsets[s]; n < call.support_offsets[s + 1]; ++n) {
      auto it = name_to_index.find(call.support_names[n]);
      if (it != name_to_index.end()) {
        idx.insert(it->second);
        name_to_index.erase(it);
      }
    }
    partition.insert(idx);
  }
  std::set<int> ref_idx;
  for (const auto& kv : name_to_index) ref_idx.insert(kv.second);
  partition.insert(ref_idx);

  std::mt19937_64 gen(seed);
  auto reservoir = [&gen](const std::set<int>& population, size_t sample_size_in) {
    const int sample_size = static_cast<int>(sample_size_in);   // ReservoirSample(int sample_size, ...)
    if (population.size() < static_cast<size_t>(sample_size)) return population;
    std::vector<int> sampled(static_cast<size_t>(sample_size));
    size_t index = 0;
    auto it = population.begin();
    for (; index < static_cast<size_t>(sample_size); ++it, ++index) sampled[index] = *it;
    for (; it != population.end(); ++it, ++index) {
      const size_t swap_index = dvo_absl::UniformClosed64(gen, 0, index);
      if (swap_index < static_cast<size_t>(sample_size)) sampled[swap_index] = *it;
    }
    return std::set<int>(sampled.begin(), sampled.end());
  };
  std::set<int> sampled, unsampled;
  for (const std::set<int>& part : partition) {
    const std::set<int> chosen = reservoir(part, static_cast<size_t>(min_per_allele));
    for (int e : part) {
      if (!chosen.count(e)) unsampled.insert(e);
    }
    sampled.insert(chosen.begin(), chosen.end());
  }
  const int remaining = max_reads - static_cast<int>(sampled.size());
  if (remaining < 0) return false;
  const std::set<int> rest = reservoir(unsampled, static_cast<size_t>(remaining));
  sampled.insert(rest.begin(), rest.end());
  out->assign(sampled.begin(), sampled.end());
  return true;
}

struct PileupRow {
  int hap_idx;
  int allele_group;
  int pos;
  int read_index;
  Row row;
};

// SortImageRows, pileup_image_native.cc:75-102.
bool SortImageRows(const dvo_read* reads, const PileupRow& a,
                   const PileupRow& b) {
  if (a.hap_idx != b.hap_idx) return a.hap_idx < b.hap_idx;
  if (a.allele_group != b.allele_group) return a.allele_group < b.allele_group;
  if (a.pos != b.pos) return a.pos < b.pos;
  const dvo_read& r1 = reads[a.read_index];
  const dvo_read& r2 = reads[b.read_index];
  return std::tuple<std::string, int>(r1.fragment_name, r1.read_number) <
         std::tuple<std::string, int>(r2.fragment_name, r2.read_number);
}

// BuildPileupForOneSample, pileup_image_native.cc:297-447.
int BuildPileup(const dvo_options& opt, const dvo_call& call,
                const std::string& ref_bases, const dvo_read* reads,
                int n_reads, int image_start_pos,
                const char* const* alt_alleles, int n_alt_alleles,
                int pileup_height, float mean_coverage,
                const int64_t* alignment_positions, const int32_t* blank,
                int n_blank, std::vector<Row>* rows_out,
                std::vector<int>* row_read_out) {
  if (static_cast<int>(ref_bases.size()) != opt.width) {
    return fail("ref_bases.size() != width");
  }
  if (pileup_height == 0) pileup_height = opt.height;
  const int max_reads = pileup_height - opt.reference_band_height;

  std::vector<Row> rows;
  std::vector<int> row_read;
  for (int i = 0; i < opt.reference_band_height; i++) {
    Row r;
    if (EncodeReference(opt, ref_bases, &r) != 0) return -1;
    rows.push_back(std::move(r));
    row_read.push_back(-1);
  }

  std::vector<int> sampled;
  if (!opt.use_non_uniform_downsampling ||
      !DownsampleWithMinsPerAllele(call, reads, n_reads, max_reads, opt.non_uniform_downsampling_threshold,
                                   opt.random_seed, &sampled)) {
    sampled = DownsampleReadIndices(n_reads, max_reads, opt.random_seed);
  }

  // read name -> allele group (pileup_image_native.cc:346-361)
  std::map<std::string, int> read_name_to_allele_group;
  int num_alt_alleles_in_variant = 0;
  if (opt.sort_by_alt_allele_support) {
    num_alt_alleles_in_variant = call.n_alts;
    for (int i = 0; i < call.n_alts; ++i) {
      for (int s = 0; s < call.n_support; ++s) {
        if (std::string(call.alts[i]) == call.support_alleles[s]) {
          for (int n = call.support_offsets[s]; n < call.support_offsets[s + 1];
               ++n) {
            read_name_to_allele_group[call.support_names[n]] = i;
          }
        }
      }
    }
  }

  std::vector<PileupRow> pileup_of_reads;
  for (int index : sampled) {
    if (static_cast<int>(pileup_of_reads.size()) >= max_reads) break;
    const dvo_read& read = reads[index];
    Row row;
    int rc = EncodeRead(opt, call, ref_bases, read, image_start_pos,
                        alt_alleles, n_alt_alleles, blank, n_blank, &row);
    if (rc < 0) return -1;
    if (rc == 0) continue;
    int hap_idx = GetHapIndex(opt, read);
    int allele_group = 0;
    if (opt.sort_by_alt_allele_support) {
      allele_group = num_alt_alleles_in_variant;
      auto it = read_name_to_allele_group.find(ReadKey(read));
      if (it != read_name_to_allele_group.end()) allele_group = it->second;
    }
    int64_t pos = alignment_positions == nullptr ? read.position
                                                 : alignment_positions[index];
    pileup_of_reads.push_back(
        {hap_idx, allele_group, static_cast<int>(pos), index, std::move(row)});
  }

  std::stable_sort(pileup_of_reads.begin(), pileup_of_reads.end(),
                   [&](const PileupRow& a, const PileupRow& b) {
                     return SortImageRows(reads, a, b);
                   });
  for (auto& pr : pileup_of_reads) {
    rows.push_back(std::move(pr.row));
    row_read.push_back(pr.read_index);
  }
  const int n_read_rows = static_cast<int>(pileup_of_reads.size());

  // blank rows (pileup_image_native.cc:412-421)
  while (static_cast<int>(rows.size()) < pileup_height) {
    rows.push_back(
        Row(opt.n_channels, std::vector<unsigned char>(ref_bases.size(), 0)));
    row_read.push_back(-1);
  }

  // mean coverage (pileup_image_native.cc:423-444)
  for (int c = 0; c < opt.n_channels; ++c) {
    if (opt.channels[c] == DVO_CH_MEAN_COVERAGE) {
      const int limit = std::min(
          static_cast<int>(mean_coverage) + opt.reference_band_height,
          pileup_height);
      for (int i = 0; i < limit; i++) {
        rows[i][c].assign(ref_bases.size(), i < opt.reference_band_height
                                                ? kChannelValue255
                                                : kChannelValue200);
      }
      break;
    }
  }
  *rows_out = std::move(rows);
  *row_read_out = std::move(row_read);
  return n_read_rows;
}

// FillPileupArray with AltAlignedPileup::kNone, pileup_image_native.h:214-275:
// channel_data[C][W] rows -> out[((row * W) + col) * c_total + c].
void FillPileupArray(const std::vector<Row>& rows, int c_total, uint8_t* out) {
  size_t pos = 0;
  for (const Row& row : rows) {
    const size_t w = row.empty() ? 0 : row[0].size();
    for (size_t col = 0; col < w; ++col) {
      size_t c = 0;
      for (; c < row.size(); ++c) out[pos++] = row[c][col];
      for (; c < static_cast<size_t>(c_total); ++c) out[pos++] = 0;
    }
  }
}

int ReadOverlaps(const dvo_read& read, int64_t start, int64_t end) {
  // nucleus/util/utils.cc:172-240
  int64_t read_end = read.position;
  for (int i = 0; i < read.n_cigar; ++i) {
    switch (read.cigar_ops[i]) {
      case DVO_CIGAR_ALIGNMENT_MATCH:
      case DVO_CIGAR_SEQUENCE_MATCH:
      case DVO_CIGAR_DELETE:
      case DVO_CIGAR_SKIP:
      case DVO_CIGAR_SEQUENCE_MISMATCH:
        read_end += read.cigar_lens[i];
        break;
      default:
        break;
    }
  }
  return end > read.position && start < read_end;
}

}  // namespace

extern "C" {

const char* dvo_last_error(void) { return g_error.c_str(); }

int dvo_channel_str_to_enum(const char* name) {
  // pileup_channel_lib.cc:421-512 and pileup_channel_lib.h:60-101
  static const std::map<std::string, int> kMap = {
      {"read_base", DVO_CH_READ_BASE},
      {"base_quality", DVO_CH_BASE_QUALITY},
      {"mapping_quality", DVO_CH_MAPPING_QUALITY},
      {"strand", DVO_CH_STRAND},
      {"read_supports_variant", DVO_CH_READ_SUPPORTS_VARIANT},
      {"read_supports_variant_fuzzy", DVO_CH_READ_SUPPORTS_VARIANT_FUZZY},
      {"base_differs_from_ref", DVO_CH_BASE_DIFFERS_FROM_REF},
      {"read_mapping_percent", DVO_CH_READ_MAPPING_PERCENT},
      {"haplotype", DVO_CH_HAPLOTYPE_TAG},
      {"allele_frequency", DVO_CH_ALLELE_FREQUENCY},
      {"diff_channels_alternate_allele_1", DVO_CH_UNSPECIFIED},
      {"diff_channels_alternate_allele_2", DVO_CH_UNSPECIFIED},
      {"avg_base_quality", DVO_CH_AVG_BASE_QUALITY},
      {"identity", DVO_CH_IDENTITY},
      {"gap_compressed_identity", DVO_CH_GAP_COMPRESSED_IDENTITY},
      {"gc_content", DVO_CH_GC_CONTENT},
      {"is_homopolymer", DVO_CH_IS_HOMOPOLYMER},
      {"homopolymer_weighted", DVO_CH_HOMOPOLYMER_WEIGHTED},
      {"blank", DVO_CH_BLANK},
      {"insert_size", DVO_CH_INSERT_SIZE},
      {"base_channels_alternate_allele_1", DVO_CH_UNSPECIFIED},
      {"base_channels_alternate_allele_2", DVO_CH_UNSPECIFIED},
      {"mean_coverage", DVO_CH_MEAN_COVERAGE},
      {"base_methylation", DVO_CH_BASE_METHYLATION},
      {"base_6ma", DVO_CH_BASE_6MA},
      {"supplementary_alignment", DVO_CH_SUPPLEMENTARY_ALIGNMENT},
      {"allele_sample_probability", DVO_CH_ALLELE_SAMPLE_PROBABILITY},
      {"homopolymer_insertion_quality", DVO_CH_HOMOPOLYMER_INSERTION_QUALITY},
      {"homopolymer_deletion_quality", DVO_CH_HOMOPOLYMER_DELETION_QUALITY},
      {"inter_homopolymer_insertion_quality", DVO_CH_INTER_HOMOPOLYMER_INSERTION_QUALITY},
  };
  auto it = kMap.find(name);
  return it == kMap.end() ? -1 : it->second;
}

int dvo_encode_reference(const dvo_options* opt, const char* ref_bases, int w,
                         uint8_t* out_hwc) {
  Row row;
  if (EncodeReference(*opt, std::string(ref_bases, w), &row) != 0) return -1;
  FillPileupArray({row}, opt->n_channels, out_hwc);
  return 0;
}

int dvo_encode_read(const dvo_options* opt, const dvo_call* call,
                    const char* ref_bases, int w, const dvo_read* read,
                    int32_t image_start_pos, const char* const* alt_alleles,
                    int n_alt_alleles, const int32_t* channels_to_blank,
                    int n_blank, uint8_t* out_hwc) {
  Row row;
  int rc = EncodeRead(*opt, *call, std::string(ref_bases, w), *read,
                      image_start_pos, alt_alleles, n_alt_alleles,
                      channels_to_blank, n_blank, &row);
  if (rc != 1) return rc;
  FillPileupArray({row}, opt->n_channels, out_hwc);
  return 1;
}

int dvo_build_pileup(const dvo_options* opt, const dvo_call* call,
                     const char* ref_bases, int w, const dvo_read* reads,
                     int n_reads, int32_t image_start_pos,
                     const char* const* alt_alleles, int n_alt_alleles,
                     int pileup_height, float mean_coverage,
                     const int64_t* alignment_positions,
                     const int32_t* channels_to_blank, int n_blank,
                     uint8_t* out_hwc, int32_t* out_row_read) {
  std::vector<Row> rows;
  std::vector<int> row_read;
  int n = BuildPileup(*opt, *call, std::string(ref_bases, w), reads, n_reads,
                      image_start_pos, alt_alleles, n_alt_alleles,
                      pileup_height, mean_coverage, alignment_positions,
                      channels_to_blank, n_blank, &rows, &row_read);
  if (n < 0) return n;
  FillPileupArray(rows, opt->n_channels, out_hwc);
  if (out_row_read != nullptr) {
    for (size_t i = 0; i < row_read.size(); ++i) out_row_read[i] = row_read[i];
  }
  return n;
}

int dvo_fuzzy_read_supports_alt(const dvo_call* call, const dvo_read* read,
                                const char* const* alt_alleles, int n_alt_alleles) {
  dvo_options opt;
  memset(&opt, 0, sizeof(opt));
  ReadChannels rc{opt, call, read, alt_alleles, n_alt_alleles};
  const int rs = rc.FuzzyReadSupportsAlt();
  if (rc.error) return fail("CHECK_LT(image_alt_allele_global_index, alt_allele_phases.size())");
  return rs;
}

int dvo_downsample_indices(int n, int max_reads, uint32_t seed, int32_t* out) {
  std::vector<int> idx = DownsampleReadIndices(n, max_reads, seed);
  for (int i = 0; i < n; ++i) out[i] = idx[i];
  return 0;
}

int dvo_read_overlaps(const dvo_read* read, int64_t start, int64_t end) {
  return ReadOverlaps(*read, start, end);
}

// Re-expand one packed item into proto-shaped inputs (oracle/packed_adapter.h) and run BuildPileup.
static int EncodePackedItem(const dvo_options& opt, const dvo_packed_batch& b,
                            int item, int out_channels, uint8_t* out,
                            int32_t* out_rows) {
  std::string error;
  const int kept = dvo_adapter::ExpandPackedItem(
      opt, b, item, &error,
      [&](const dvo_call& call, const std::string& ref, const dvo_read* reads, int n, int image_start,
          const char* const* alt_alleles, int n_alt_alleles, int h, float mean_cov, const int64_t* sort_pos,
          const int32_t* blank, int n_blank) {
        std::vector<Row> rows;
        std::vector<int> row_read;
        const int k = BuildPileup(opt, call, ref, reads, n, image_start, alt_alleles, n_alt_alleles, h, mean_cov,
                                  sort_pos, blank, n_blank, &rows, &row_read);
        if (k >= 0) FillPileupArray(rows, out_channels, out + b.item_out_off[item]);
        return k;
      });
  if (kept < 0) {
    if (!error.empty()) g_error = error;
    return -1;
  }
  if (out_rows) out_rows[item] = kept;
  return 0;
}

int dvo_encode_packed(const dvo_options* opt, const dvo_packed_batch* b,
                      int out_channels, uint8_t* out, int32_t* out_rows,
                      int n_threads) {
  if (out_channels < opt->n_channels) return fail("out_channels too small");
  if (n_threads <= 1) {
    for (int i = 0; i < b->n_items; ++i) {
      if (EncodePackedItem(*opt, *b, i, out_channels, out, out_rows) != 0)
        return -1;
    }
    return 0;
  }
  std::vector<std::thread> threads;
  std::vector<int> status(n_threads, 0);
  for (int t = 0; t < n_threads; ++t) {
    threads.emplace_back([&, t]() {
      for (int i = t; i < b->n_items; i += n_threads) {
        if (EncodePackedItem(*opt, *b, i, out_channels, out, out_rows) != 0) {
          status[t] = -1;
          return;
        }
      }
    });
  }
  for (auto& th : threads) th.join();
  for (int s : status) {
    if (s != 0) return fail("dvo_encode_packed: worker failed");
  }
  return 0;
}

}  // extern "C"
