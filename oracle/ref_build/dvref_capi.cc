// dvref_capi.cc -- the REFERENCE's own pileup encoder behind the oracle's C interface (oracle/dvo.h).
//
// TEST INFRASTRUCTURE.  oracle/ref_build/Makefile compiles, from where they lie under /root/reference and
// UNMODIFIED, deepvariant/pileup_image_native.cc, deepvariant/pileup_channel_lib.cc and deepvariant/channels/*.cc
// (against headers mini_protoc.py generates from the reference's .proto files and the small abseil / protobuf
// stand-ins under shims/) and links them with this file into oracle/_ref/libdvref.so.  This file is the only
// code of ours in that library: it turns the proto-shaped C inputs of dvo.h into the message objects the
// reference's classes take and copies their ImageRows out with the reference's own FillPileupArray.
//
// The library exports the SAME symbols as libdvoracle.so (dvo_*), so oracle/oracle.py drives either with the
// same ctypes code: tests/test_reference_encoder_cpu.py holds the restatement (encoder_oracle.cpp) against the
// reference itself on the known-answer vectors, the golden pileups and seeded fuzz inputs, and bench.py's
// cpu_baseline times this library ("kind": "reference").
//
// Differences from a bazel build of the same sources, all outside what is compared: LOG(FATAL) / CHECK failures
// throw (returned as an error code) instead of aborting; google::protobuf::Map iterates in key order;
// absl::Uniform (non-uniform downsampling only) draws from the standard library.
#include <cstring>
#include <memory>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "deepvariant/channels/read_supports_variant_fuzzy_channel.h"
#include "deepvariant/pileup_channel_lib.h"
#include "deepvariant/pileup_image_native.h"
#include "deepvariant/protos/deepvariant.pb.h"
#include "dvo.h"
#include "packed_adapter.h"
#include "third_party/nucleus/protos/range.pb.h"
#include "third_party/nucleus/protos/reads.pb.h"
#include "third_party/nucleus/util/utils.h"

namespace learning {
namespace genomics {
namespace deepvariant {
// defined in pileup_image_native.cc:153 (not declared in its header)
std::vector<int> DownsampleReadIndices(const std::vector<const ::nucleus::genomics::v1::Read*>& reads, int max_reads,
                                       std::mt19937_64 gen);
}  // namespace deepvariant
}  // namespace genomics
}  // namespace learning

namespace {

namespace refdv = learning::genomics::deepvariant;
using nucleus::genomics::v1::CigarUnit;
using nucleus::genomics::v1::Read;

thread_local std::string g_error;

int fail(const std::string& m) {
  g_error = m;
  return -1;
}

// channel names the reference's ChannelStrToEnum knows (pileup_channel_lib.h:60-101), for the way back from
// dvo_options.channels (enum values) to PileupImageOptions.channels (names)
const char* const kChannelNames[] = {
    "read_base", "base_quality", "mapping_quality", "strand", "read_supports_variant", "read_supports_variant_fuzzy",
    "base_differs_from_ref", "read_mapping_percent", "haplotype", "allele_frequency", "avg_base_quality", "identity",
    "gap_compressed_identity", "gc_content", "is_homopolymer", "homopolymer_weighted", "blank", "insert_size",
    "mean_coverage", "base_methylation", "base_6ma", "supplementary_alignment", "allele_sample_probability",
    "homopolymer_insertion_quality", "homopolymer_deletion_quality", "inter_homopolymer_insertion_quality",
};

const char* ChannelName(int channel_enum) {
  for (const char* name : kChannelNames) {
    if (static_cast<int>(refdv::Channels::ChannelStrToEnum(name)) == channel_enum) return name;
  }
  return nullptr;
}

bool MakeOptions(const dvo_options& o, refdv::PileupImageOptions* p) {
  p->set_width(o.width);
  p->set_height(o.height);
  p->set_reference_band_height(o.reference_band_height);
  p->set_base_color_offset_a_and_g(o.base_color_offset_a_and_g);
  p->set_base_color_offset_t_and_c(o.base_color_offset_t_and_c);
  p->set_base_color_stride(o.base_color_stride);
  p->set_allele_supporting_read_alpha(o.allele_supporting_read_alpha);
  p->set_allele_unsupporting_read_alpha(o.allele_unsupporting_read_alpha);
  p->set_other_allele_supporting_read_alpha(o.other_allele_supporting_read_alpha);
  p->set_reference_matching_read_alpha(o.reference_matching_read_alpha);
  p->set_reference_mismatching_read_alpha(o.reference_mismatching_read_alpha);
  if (o.indel_anchoring_base_char) p->set_indel_anchoring_base_char(std::string(1, static_cast<char>(o.indel_anchoring_base_char)));
  p->set_reference_base_quality(o.reference_base_quality);
  p->set_positive_strand_color(o.positive_strand_color);
  p->set_negative_strand_color(o.negative_strand_color);
  p->set_base_quality_cap(o.base_quality_cap);
  p->set_mapping_quality_cap(o.mapping_quality_cap);
  p->mutable_read_requirements()->set_min_base_quality(o.min_base_quality);
  p->mutable_read_requirements()->set_min_mapping_quality(o.min_mapping_quality);
  p->set_random_seed(o.random_seed);
  p->set_sort_by_haplotypes(o.sort_by_haplotypes != 0);
  p->set_hp_tag_for_assembly_polishing(o.hp_tag_for_assembly_polishing);
  p->set_sort_by_alt_allele_support(o.sort_by_alt_allele_support != 0);
  p->set_min_non_zero_allele_frequency(o.min_non_zero_allele_frequency);
  p->set_num_channels(o.n_channels);
  for (int c = 0; c < o.n_channels; ++c) {
    const char* name = ChannelName(o.channels[c]);
    if (!name) return false;
    p->add_channels(name);
  }
  return true;
}

void MakeRead(const dvo_read& r, Read* read) {
  read->set_fragment_name(r.fragment_name ? r.fragment_name : "");
  read->set_read_number(r.read_number);
  auto* aln = read->mutable_alignment();
  aln->mutable_position()->set_reference_name("contig");
  aln->mutable_position()->set_position(r.position);
  aln->mutable_position()->set_reverse_strand(r.reverse_strand != 0);
  aln->set_mapping_quality(r.mapping_quality);
  for (int i = 0; i < r.n_cigar; ++i) {
    auto* cu = aln->add_cigar();
    cu->set_operation(static_cast<CigarUnit::Operation>(r.cigar_ops[i]));
    cu->set_operation_length(r.cigar_lens[i]);
  }
  read->set_supplementary_alignment(r.supplementary != 0);
  read->set_fragment_length(r.fragment_length);
  read->set_aligned_sequence(std::string(r.seq ? r.seq : "", static_cast<size_t>(r.seq_len)));
  read->set_aligned_quality(std::string(reinterpret_cast<const char*>(r.qual), static_cast<size_t>(r.qual_len)));
  if (r.hp_present) {
    auto& hp = (*read->mutable_info())["HP"];
    for (int i = 0; i < r.hp_n_values; ++i) {
      auto* v = hp.add_values();
      if (i == 0 && !r.hp_is_int) {
        v->set_string_value("1");
      } else {
        v->set_int_value(i == 0 ? r.hp_value : 0);
      }
    }
  }
  if (r.tp) {
    auto& tp = (*read->mutable_info())["tp"];
    for (int i = 0; i < r.tp_len; ++i) tp.add_values()->set_int_value(r.tp[i]);
  }
  if (r.t0) {
    (*read->mutable_info())["t0"].add_values()->set_string_value(std::string(r.t0, static_cast<size_t>(r.t0_len)));
  }
  if (r.mod_5mc) {
    (*read->mutable_base_modifications())["5mC"] = std::string(reinterpret_cast<const char*>(r.mod_5mc), static_cast<size_t>(r.mod_5mc_len));
  }
  if (r.mod_6ma) {
    (*read->mutable_base_modifications())["6mA"] = std::string(reinterpret_cast<const char*>(r.mod_6ma), static_cast<size_t>(r.mod_6ma_len));
  }
}

void MakeCall(const dvo_call& c, refdv::DeepVariantCall* call) {
  auto* v = call->mutable_variant();
  v->set_start(c.variant_start);
  if (c.reference_bases) v->set_reference_bases(c.reference_bases);
  for (int i = 0; i < c.n_alts; ++i) v->add_alternate_bases(c.alts[i]);
  for (int i = 0; i < c.n_rejected_alts; ++i) v->add_alternate_bases_rejected(c.rejected_alts[i]);
  for (int s = 0; s < c.n_support; ++s) {
    auto& sr = (*call->mutable_allele_support())[c.support_alleles[s]];
    for (int n = c.support_offsets[s]; n < c.support_offsets[s + 1]; ++n) sr.add_read_names(c.support_names[n]);
  }
  for (int s = 0; s < c.n_rejected_support; ++s) {
    auto& sr = (*call->mutable_rejected_allele_support())[c.rejected_support_alleles[s]];
    for (int n = c.rejected_support_offsets[s]; n < c.rejected_support_offsets[s + 1]; ++n) {
      sr.add_read_names(c.rejected_support_names[n]);
    }
  }
  for (int i = 0; i < c.n_af; ++i) (*call->mutable_allele_frequency())[c.af_alleles[i]] = c.af_values[i];
  for (int i = 0; i < c.n_ref_support; ++i) {
    call->add_ref_support(c.ref_support_names ? c.ref_support_names[i] : "");
  }
  if (c.alt_ps_present) {
    auto& ps = (*v->mutable_info())["ALT_PS"];
    for (int i = 0; i < c.n_alt_ps; ++i) ps.add_values()->set_int_value(c.alt_ps[i]);
  }
}

absl::flat_hash_set<refdv::DeepVariantChannelEnum> BlankSet(const int32_t* blank, int n) {
  absl::flat_hash_set<refdv::DeepVariantChannelEnum> out;
  for (int i = 0; i < n; ++i) out.insert(static_cast<refdv::DeepVariantChannelEnum>(blank[i]));
  return out;
}

std::vector<std::string> Strings(const char* const* a, int n) {
  std::vector<std::string> out;
  for (int i = 0; i < n; ++i) out.emplace_back(a[i]);
  return out;
}

// FillPileupArray of the reference (pileup_image_native.h:214-335, AltAlignedPileup::kNone), then widened to
// c_total channels per pixel when the caller's tensor has more (the oracle pads the same way).
void CopyOut(const std::vector<std::unique_ptr<refdv::ImageRow>>& rows, int n_channels, int c_total, uint8_t* out) {
  if (rows.empty()) return;
  const size_t w = static_cast<size_t>(rows[0]->Width());
  std::vector<uint8_t> dense(rows.size() * w * static_cast<size_t>(n_channels));
  refdv::FillPileupArray(absl::MakeConstSpan(rows), absl::Span<const std::vector<std::unique_ptr<refdv::ImageRow>>>(),
                      refdv::AltAlignedPileup::kNone, &dense, static_cast<int>(dense.size()), 0);
  if (c_total == n_channels) {
    std::memcpy(out, dense.data(), dense.size());
    return;
  }
  size_t src = 0, dst = 0;
  for (size_t px = 0; px < rows.size() * w; ++px) {
    for (int c = 0; c < n_channels; ++c) out[dst++] = dense[src++];
    for (int c = n_channels; c < c_total; ++c) out[dst++] = 0;
  }
}

int BuildPileup(const dvo_options& opt, const dvo_call& call, const std::string& ref_bases, const dvo_read* reads,
                int n_reads, int image_start_pos, const char* const* alt_alleles, int n_alt_alleles, int pileup_height,
                float mean_coverage, const int64_t* alignment_positions, const int32_t* blank, int n_blank, int c_total,
                uint8_t* out, int32_t* row_read = nullptr) {
  refdv::PileupImageOptions options;
  if (!MakeOptions(opt, &options)) return fail("a channel of dvo_options has no name the reference knows");
  refdv::PileupImageEncoderNative encoder(options);
  refdv::DeepVariantCall dv_call;
  MakeCall(call, &dv_call);
  std::vector<Read> protos(static_cast<size_t>(n_reads));
  std::vector<const Read*> ptrs;
  for (int i = 0; i < n_reads; ++i) {
    MakeRead(reads[i], &protos[static_cast<size_t>(i)]);
    ptrs.push_back(&protos[static_cast<size_t>(i)]);
  }
  refdv::SampleOptions sample;
  sample.set_pileup_height(pileup_height);
  sample.set_use_non_uniform_downsampling(opt.use_non_uniform_downsampling != 0);
  sample.set_non_uniform_downsampling_threshold(opt.non_uniform_downsampling_threshold);
  std::vector<int64_t> positions;
  if (alignment_positions) positions.assign(alignment_positions, alignment_positions + n_reads);
  auto rows = encoder.BuildPileupForOneSample(dv_call, ref_bases, ptrs, image_start_pos, Strings(alt_alleles, n_alt_alleles),
                                              sample, mean_coverage, alignment_positions ? &positions : nullptr,
                                              BlankSet(blank, n_blank));
  CopyOut(rows, opt.n_channels, c_total, out);
  // The reference does not say how many reads it drew: rows below the band that hold any nonzero byte
  int kept = 0;
  const size_t row_bytes = ref_bases.size() * static_cast<size_t>(c_total);
  for (size_t r = static_cast<size_t>(opt.reference_band_height); r < rows.size(); ++r) {
    bool any = false;
    for (size_t k = 0; k < row_bytes && !any; ++k) any = out[r * row_bytes + k] != 0;
    if (row_read) row_read[r] = any ? -2 : -1;
    kept += any ? 1 : 0;
  }
  if (row_read) {
    for (int r = 0; r < opt.reference_band_height && r < static_cast<int>(rows.size()); ++r) row_read[r] = -1;
  }
  return kept;
}

template <class F>
int Guard(F f) {
  try {
    return f();
  } catch (const std::exception& e) {
    return fail(e.what());
  }
}

}  // namespace

extern "C" {

const char* dvo_last_error(void) { return g_error.c_str(); }

int dvo_is_reference(void) { return 1; }

int dvo_channel_str_to_enum(const char* name) {
  return Guard([&] { return static_cast<int>(refdv::Channels::ChannelStrToEnum(name)); });
}

int dvo_encode_reference(const dvo_options* opt, const char* ref_bases, int w, uint8_t* out_hwc) {
  return Guard([&] {
    refdv::PileupImageOptions options;
    if (!MakeOptions(*opt, &options)) return fail("a channel of dvo_options has no name the reference knows");
    refdv::PileupImageEncoderNative encoder(options);
    std::vector<std::unique_ptr<refdv::ImageRow>> rows;
    rows.push_back(encoder.EncodeReference(std::string(ref_bases, static_cast<size_t>(w))));
    CopyOut(rows, opt->n_channels, opt->n_channels, out_hwc);
    return 0;
  });
}

int dvo_encode_read(const dvo_options* opt, const dvo_call* call, const char* ref_bases, int w, const dvo_read* read,
                    int32_t image_start_pos, const char* const* alt_alleles, int n_alt_alleles,
                    const int32_t* channels_to_blank, int n_blank, uint8_t* out_hwc) {
  return Guard([&] {
    refdv::PileupImageOptions options;
    if (!MakeOptions(*opt, &options)) return fail("a channel of dvo_options has no name the reference knows");
    refdv::PileupImageEncoderNative encoder(options);
    refdv::DeepVariantCall dv_call;
    MakeCall(*call, &dv_call);
    Read proto;
    MakeRead(*read, &proto);
    std::vector<std::unique_ptr<refdv::ImageRow>> rows;
    rows.push_back(encoder.EncodeRead(dv_call, std::string(ref_bases, static_cast<size_t>(w)), proto, image_start_pos,
                                      Strings(alt_alleles, n_alt_alleles), BlankSet(channels_to_blank, n_blank)));
    if (!rows[0]) return 0;
    CopyOut(rows, opt->n_channels, opt->n_channels, out_hwc);
    return 1;
  });
}

/* out_row_read: the reference does not say which read a row shows; read rows come back as -2, the others -1,
 * and the return value counts the rows below the band that hold any nonzero byte. */
int dvo_build_pileup(const dvo_options* opt, const dvo_call* call, const char* ref_bases, int w, const dvo_read* reads,
                     int n_reads, int32_t image_start_pos, const char* const* alt_alleles, int n_alt_alleles,
                     int pileup_height, float mean_coverage, const int64_t* alignment_positions,
                     const int32_t* channels_to_blank, int n_blank, uint8_t* out_hwc, int32_t* out_row_read) {
  return Guard([&] {
    return BuildPileup(*opt, *call, std::string(ref_bases, static_cast<size_t>(w)), reads, n_reads, image_start_pos,
                       alt_alleles, n_alt_alleles, pileup_height, mean_coverage, alignment_positions, channels_to_blank,
                       n_blank, opt->n_channels, out_hwc, out_row_read);
  });
}

int dvo_fuzzy_read_supports_alt(const dvo_call* call, const dvo_read* read, const char* const* alt_alleles,
                                int n_alt_alleles) {
  return Guard([&] {
    refdv::PileupImageOptions options;
    options.set_width(221);
    refdv::ReadSupportsVariantFuzzyChannel channel(221, options);
    refdv::DeepVariantCall dv_call;
    MakeCall(*call, &dv_call);
    Read proto;
    MakeRead(*read, &proto);
    const std::vector<std::string> alts = Strings(alt_alleles, n_alt_alleles);
    return channel.ReadSupportsAlt(dv_call, proto, alts);
  });
}

int dvo_downsample_indices(int n, int max_reads, uint32_t seed, int32_t* out) {
  return Guard([&] {
    std::vector<const Read*> reads(static_cast<size_t>(n), nullptr);   // only the count is looked at
    const std::vector<int> idx = refdv::DownsampleReadIndices(reads, max_reads, std::mt19937_64(seed));
    for (int i = 0; i < n; ++i) out[i] = idx[static_cast<size_t>(i)];
    return 0;
  });
}

/* nucleus::ReadOverlapsRegion (third_party/nucleus/util/utils.cc:172-240) on one contig. */
int dvo_read_overlaps(const dvo_read* read, int64_t start, int64_t end) {
  return Guard([&] {
    Read proto;
    MakeRead(*read, &proto);
    nucleus::genomics::v1::Range range;
    range.set_reference_name("contig");
    range.set_start(start);
    range.set_end(end);
    return nucleus::ReadOverlapsRegion(proto, range) ? 1 : 0;
  });
}

int dvo_encode_packed(const dvo_options* opt, const dvo_packed_batch* b, int out_channels, uint8_t* out,
                      int32_t* out_rows, int n_threads) {
  if (out_channels < opt->n_channels) return fail("out_channels too small");
  auto one = [&](int item) -> int {
    return Guard([&] {
      std::string error;
      const int rc = dvo_adapter::ExpandPackedItem(
          *opt, *b, item, &error,
          [&](const dvo_call& call, const std::string& ref, const dvo_read* reads, int n, int image_start,
              const char* const* alt_alleles, int n_alt_alleles, int h, float mean_cov, const int64_t* sort_pos,
              const int32_t* blank, int n_blank) {
            return BuildPileup(*opt, call, ref, reads, n, image_start, alt_alleles, n_alt_alleles, h, mean_cov, sort_pos,
                               blank, n_blank, out_channels, out + b->item_out_off[item]);
          });
      if (rc < 0 && !error.empty()) return fail(error);
      if (rc >= 0 && out_rows) out_rows[item] = rc;
      return rc < 0 ? -1 : 0;
    });
  };
  if (n_threads <= 1) {
    for (int i = 0; i < b->n_items; ++i) {
      if (one(i) != 0) return -1;
    }
    return 0;
  }
  std::vector<std::thread> threads;
  std::vector<std::string> errors(static_cast<size_t>(n_threads));
  for (int t = 0; t < n_threads; ++t) {
    threads.emplace_back([&, t]() {
      for (int i = t; i < b->n_items; i += n_threads) {
        if (one(i) != 0) {
          errors[static_cast<size_t>(t)] = g_error.empty() ? "worker failed" : g_error;
          return;
        }
      }
    });
  }
  for (auto& th : threads) th.join();
  for (const std::string& e : errors) {
    if (!e.empty()) return fail("dvo_encode_packed (reference): " + e);
  }
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Candidate generation: the reference's AlleleCounter (deepvariant/allelecounter.cc) and its multi-sample
// VariantCaller (deepvariant/variant_calling_multisample.cc, what make_examples runs -- with ONE sample here),
// compiled unmodified like the encoder above.  SURVEY.md 8f row f2.  One call counts a region and, if asked,
// calls its candidates; results come back as tab-separated text (oracle/oracle.py parses it).
// ---------------------------------------------------------------------------------------------------------------
#include <sstream>
#include <unordered_map>

#include "deepvariant/allelecounter.h"
#include "deepvariant/variant_calling_multisample.h"
#include "third_party/nucleus/io/reference.h"

namespace {

// nucleus::GenomeReference over one stretch of one contig handed in by the test
class WindowReference : public nucleus::GenomeReference {
 public:
  WindowReference(const std::string& contig, int64_t contig_length, int64_t start, std::string bases)
      : start_(start), bases_(std::move(bases)) {
    contigs_.emplace_back();
    contigs_.back().set_name(contig);
    contigs_.back().set_n_bases(contig_length);
  }
  const std::vector<nucleus::genomics::v1::ContigInfo>& Contigs() const override { return contigs_; }
  nucleus::StatusOr<std::string> GetBases(const nucleus::genomics::v1::Range& range) const override {
    if (!IsValidInterval(range)) return nucleus::InvalidArgument("Invalid interval");
    if (range.start() < start_ || range.end() > start_ + static_cast<int64_t>(bases_.size())) {
      return nucleus::OutOfRange("the test handed in no reference bases for this interval");
    }
    return bases_.substr(static_cast<size_t>(range.start() - start_), static_cast<size_t>(range.end() - range.start()));
  }

 private:
  int64_t start_;
  std::string bases_;
  std::vector<nucleus::genomics::v1::ContigInfo> contigs_;
};

std::string Join(const google::protobuf::RepeatedPtrField<std::string>& v) {
  std::string out;
  for (int i = 0; i < v.size(); ++i) {
    if (i) out += ',';
    out += v[i];
  }
  return out;
}

}  // namespace

extern "C" {

typedef struct dvr_calling_options {
  /* AlleleCounterOptions (deepvariant.proto) */
  int32_t partition_size;
  int32_t min_mapping_quality;
  int32_t min_base_quality;
  int32_t track_ref_reads;
  int32_t normalize_reads;
  int32_t keep_legacy_behavior;
  /* VariantCallerOptions; call_variants = 0 -> counts only */
  int32_t call_variants;
  int32_t min_count_snps;
  int32_t min_count_indels;
  float min_fraction_snps;
  float min_fraction_indels;
  float min_fraction_multiplier;
  float p_error;
  int32_t max_gq;
  int32_t gq_resolution;
  int32_t ploidy;
  int32_t call_positions_only; /* CallPositionsFromAlleleCounts instead of CallsFromAlleleCounts */
} dvr_calling_options;

void dvr_free(char* p) { std::free(p); }

/* Lines:  C <position> <ref_base> <ref_supporting_read_count>      one per position that holds anything
 *         A <read key> <bases> <type> <is_low_quality>             the read alleles of the position above, key order
 *         V <start> <end> <reference_bases> <alt,alt,...>          one per DeepVariantCall
 *         S <allele> <read name,read name,...>                     allele_support of the call above (key order)
 *         R <read name,...>                                        ref_support
 *         E <allele> <name:is_low_quality,...>                     allele_support_ext
 *         F <name:is_low_quality,...>                              ref_support_ext
 *         I <key> <v,v,...>                                        calls(0).info of the variant (AD, DP, VAF, ...)
 *         G <sample name> <genotype,...>                           calls(0)
 *         P <position>                                             (call_positions_only) */
int dvr_count_and_call(const char* contig, int64_t contig_length, int64_t ref_start, const char* ref_bases,
                       int64_t n_ref_bases, int64_t start, int64_t end, int64_t full_start, int64_t full_end,
                       const dvo_read* reads, int n_reads, const char* sample, const int32_t* candidate_positions,
                       int n_candidate_positions, const dvr_calling_options* o, char** out, uint64_t* out_len) {
  return Guard([&] {
    WindowReference ref(contig, contig_length, ref_start, std::string(ref_bases, static_cast<size_t>(n_ref_bases)));
    refdv::AlleleCounterOptions co;
    co.set_partition_size(o->partition_size);
    co.mutable_read_requirements()->set_min_mapping_quality(o->min_mapping_quality);
    co.mutable_read_requirements()->set_min_base_quality(o->min_base_quality);
    co.set_track_ref_reads(o->track_ref_reads != 0);
    co.set_normalize_reads(o->normalize_reads != 0);
    co.set_keep_legacy_behavior(o->keep_legacy_behavior != 0);
    nucleus::genomics::v1::Range range, full;
    range.set_reference_name(contig);
    range.set_start(start);
    range.set_end(end);
    full.set_reference_name(contig);
    full.set_start(full_start);
    full.set_end(full_end);
    const std::vector<int> positions(candidate_positions, candidate_positions + n_candidate_positions);
    std::unique_ptr<refdv::AlleleCounter> counter(
        full_end > full_start ? new refdv::AlleleCounter(&ref, range, full, positions, co)
                              : new refdv::AlleleCounter(&ref, range, positions, co));
    for (int i = 0; i < n_reads; ++i) {
      Read proto;
      MakeRead(reads[i], &proto);
      proto.mutable_alignment()->mutable_position()->set_reference_name(contig);
      if (o->normalize_reads) {
        std::unique_ptr<std::vector<CigarUnit>> norm_cigar(new std::vector<CigarUnit>());
        int shift = 0;
        counter->NormalizeAndAdd(proto, sample, norm_cigar, shift);
      } else {
        counter->Add(proto, sample);
      }
    }
    std::ostringstream text;
    for (const refdv::AlleleCount& c : counter->Counts()) {
      if (c.ref_supporting_read_count() == 0 && c.read_alleles().empty()) continue;
      text << "C\t" << c.position().position() << '\t' << c.ref_base() << '\t' << c.ref_supporting_read_count() << '\n';
      for (const auto& kv : c.read_alleles()) {
        text << "A\t" << kv.first << '\t' << kv.second.bases() << '\t' << static_cast<int>(kv.second.type()) << '\t'
             << (kv.second.is_low_quality() ? 1 : 0) << '\n';
      }
    }
    if (o->call_variants) {
      refdv::VariantCallerOptions vo;
      vo.set_min_count_snps(o->min_count_snps);
      vo.set_min_count_indels(o->min_count_indels);
      vo.set_min_fraction_snps(o->min_fraction_snps);
      vo.set_min_fraction_indels(o->min_fraction_indels);
      vo.set_min_fraction_multiplier(o->min_fraction_multiplier);
      vo.set_sample_name(sample);
      vo.set_p_error(o->p_error);
      vo.set_max_gq(o->max_gq);
      vo.set_gq_resolution(o->gq_resolution);
      vo.set_ploidy(o->ploidy);
      vo.set_track_ref_reads(o->track_ref_reads != 0);
      refdv::multi_sample::VariantCaller caller(vo);
      std::unordered_map<std::string, refdv::AlleleCounter*> counters{{sample, counter.get()}};
      if (o->call_positions_only) {
        for (int p : caller.CallPositionsFromAlleleCounts(counters, sample)) text << "P\t" << p << '\n';
      } else {
        for (const refdv::DeepVariantCall& call : caller.CallsFromAlleleCounts(counters, sample)) {
          const auto& v = call.variant();
          text << "V\t" << v.start() << '\t' << v.end() << '\t' << v.reference_bases() << '\t' << Join(v.alternate_bases()) << '\n';
          for (const auto& kv : call.allele_support()) text << "S\t" << kv.first << '\t' << Join(kv.second.read_names()) << '\n';
          if (call.ref_support_size()) text << "R\t" << Join(call.ref_support()) << '\n';
          for (const auto& kv : call.allele_support_ext()) {
            text << "E\t" << kv.first << '\t';
            for (int i = 0; i < kv.second.read_infos_size(); ++i) {
              text << (i ? "," : "") << kv.second.read_infos(i).read_name() << ':' << (kv.second.read_infos(i).is_low_quality() ? 1 : 0);
            }
            text << '\n';
          }
          if (call.ref_support_ext().read_infos_size()) {
            text << "F\t";
            for (int i = 0; i < call.ref_support_ext().read_infos_size(); ++i) {
              text << (i ? "," : "") << call.ref_support_ext().read_infos(i).read_name() << ':'
                   << (call.ref_support_ext().read_infos(i).is_low_quality() ? 1 : 0);
            }
            text << '\n';
          }
          if (v.calls_size()) {
            const auto& vc = v.calls(0);
            text << "G\t" << vc.call_set_name() << '\t';
            for (int i = 0; i < vc.genotype_size(); ++i) text << (i ? "," : "") << vc.genotype(i);
            text << '\n';
            for (const auto& kv : vc.info()) {
              text << "I\t" << kv.first << '\t';
              for (int i = 0; i < kv.second.values_size(); ++i) {
                const auto& val = kv.second.values(i);
                text << (i ? "," : "");
                if (val.kind_case() == nucleus::genomics::v1::Value::kIntValue) {
                  text << val.int_value();
                } else if (val.kind_case() == nucleus::genomics::v1::Value::kNumberValue) {
                  char buf[64];
                  std::snprintf(buf, sizeof(buf), "%.17g", val.number_value());
                  text << buf;
                } else {
                  text << val.string_value();
                }
              }
              text << '\n';
            }
          }
        }
      }
    }
    const std::string s = text.str();
    *out = static_cast<char*>(std::malloc(s.size() + 1));
    std::memcpy(*out, s.data(), s.size());
    (*out)[s.size()] = 0;
    *out_len = s.size();
    return 0;
  });
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// The realigner's window selector (deepvariant/realigner/window_selector.cc, compiled unmodified): per position
// of [start, end) the number of reads supporting a variant there (VariantReadsWindowSelectorCandidates) and the
// linear model's score (AlleleCountLinearWindowSelectorCandidates, float32), from the reference's AlleleCounter
// over the same reads.  SURVEY.md 8f row f4.
// ---------------------------------------------------------------------------------------------------------------
#include "deepvariant/protos/realigner.pb.h"
#include "deepvariant/realigner/window_selector.h"

extern "C" {

/* linear: bias, coeff_soft_clip, coeff_substitution, coeff_insertion, coeff_deletion, coeff_reference,
 * decision_boundary -- or NULL (out_scores is then left alone). */
int dvr_window_candidates(const char* contig, int64_t contig_length, int64_t ref_start, const char* ref_bases,
                          int64_t n_ref_bases, int64_t start, int64_t end, const dvo_read* reads, int n_reads,
                          int32_t min_mapq, int32_t min_base_quality, int32_t keep_legacy_behavior,
                          int32_t min_allele_support, int32_t enable_strict_insertion_filter, const float* linear,
                          int32_t* out_counts, float* out_scores) {
  return Guard([&] {
    WindowReference ref(contig, contig_length, ref_start, std::string(ref_bases, static_cast<size_t>(n_ref_bases)));
    refdv::AlleleCounterOptions co;
    co.set_partition_size(static_cast<int32_t>(end - start));
    co.mutable_read_requirements()->set_min_mapping_quality(min_mapq);
    co.mutable_read_requirements()->set_min_base_quality(min_base_quality);
    co.set_keep_legacy_behavior(keep_legacy_behavior != 0);
    nucleus::genomics::v1::Range range;
    range.set_reference_name(contig);
    range.set_start(start);
    range.set_end(end);
    refdv::AlleleCounter counter(&ref, range, {}, co);
    for (int i = 0; i < n_reads; ++i) {
      Read proto;
      MakeRead(reads[i], &proto);
      proto.mutable_alignment()->mutable_position()->set_reference_name(contig);
      counter.Add(proto, "placeholder_sample_id");
    }
    refdv::WindowSelectorOptions config;
    config.set_min_allele_support(min_allele_support);
    config.set_enable_strict_insertion_filter(enable_strict_insertion_filter != 0);
    const std::vector<int> counts = refdv::VariantReadsWindowSelectorCandidates(counter, config);
    for (size_t i = 0; i < counts.size(); ++i) out_counts[i] = counts[i];
    if (linear && out_scores) {
      refdv::WindowSelectorModel::AlleleCountLinearModel model;
      model.set_bias(linear[0]);
      model.set_coeff_soft_clip(linear[1]);
      model.set_coeff_substitution(linear[2]);
      model.set_coeff_insertion(linear[3]);
      model.set_coeff_deletion(linear[4]);
      model.set_coeff_reference(linear[5]);
      model.set_decision_boundary(linear[6]);
      const std::vector<float> scores = refdv::AlleleCountLinearWindowSelectorCandidates(counter, model);
      for (size_t i = 0; i < scores.size(); ++i) out_scores[i] = scores[i];
    }
    return static_cast<int>(counts.size());
  });
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// The read realigner (deepvariant/realigner/fast_pass_aligner.cc, ssw.cc) and the trimmed-read / alt-haplotype
// helpers (deepvariant/alt_aligned_pileup_lib.cc), compiled unmodified.  The one thing underneath them that is
// NOT the reference's is the local aligner: libssw is not vendored, so `src/ssw_cpp.h` here is the library's C++
// interface implemented on the product's restatement (deepvariant_amd/csrc/local_align.cpp) -- see
// shims/src/ssw_cpp.h.  SURVEY.md 8(a) row a16 and 8f row f4.
// ---------------------------------------------------------------------------------------------------------------
// (dvr_align_reads_state below runs the aligner's stages one by one, two of which are private members)
#define private public
#include "deepvariant/realigner/fast_pass_aligner.h"
#undef private
#include "deepvariant/alt_aligned_pileup_lib.h"

namespace {

void ReadLine(std::ostringstream& text, const Read& r) {
  // E = the empty Read the aligner returns under force_alignment when nothing was found
  if (r.aligned_sequence().empty() && r.fragment_name().empty() && r.alignment().cigar_size() == 0) {
    text << "E\n";
    return;
  }
  text << "R\t" << r.fragment_name() << '\t' << r.read_number() << '\t' << r.alignment().position().position() << '\t';
  for (int i = 0; i < r.alignment().cigar_size(); ++i) {
    text << (i ? "," : "") << static_cast<int>(r.alignment().cigar(i).operation()) << ':' << r.alignment().cigar(i).operation_length();
  }
  text << '\t' << r.aligned_sequence() << '\t';
  for (size_t i = 0; i < r.aligned_quality().size(); ++i) {
    text << (i ? "," : "") << static_cast<int>(static_cast<unsigned char>(r.aligned_quality()[i]));
  }
  text << '\t' << r.alignment().mapping_quality() << '\t' << (r.alignment().position().reverse_strand() ? 1 : 0);
  for (const char* key : {"5mC", "6mA"}) {
    text << '\t';
    auto it = r.base_modifications().find(key);
    if (it != r.base_modifications().end()) {
      for (size_t i = 0; i < it->second.size(); ++i) text << (i ? "," : "") << static_cast<int>(static_cast<unsigned char>(it->second[i]));
    } else {
      text << '-';
    }
  }
  text << '\n';
}

int TextOut(const std::ostringstream& text, char** out, uint64_t* out_len) {
  const std::string s = text.str();
  *out = static_cast<char*>(std::malloc(s.size() + 1));
  std::memcpy(*out, s.data(), s.size());
  (*out)[s.size()] = 0;
  *out_len = s.size();
  return 0;
}

void SetAlignerOptions(const int32_t* o, double similarity, refdv::AlignerOptions* a) {
  a->set_match(o[0]);
  a->set_mismatch(o[1]);
  a->set_gap_open(o[2]);
  a->set_gap_extend(o[3]);
  a->set_kmer_size(o[4]);
  a->set_read_size(o[5]);
  a->set_max_num_of_mismatches(o[6]);
  a->set_realignment_similarity_threshold(similarity);
  a->set_force_alignment(o[7] != 0);
}

}  // namespace

extern "C" {

/* FastPassAligner::AlignReads as the window realigner drives it (realigner.py:740-790).
 * options: match, mismatch, gap_open, gap_extend, kmer_size, read_size, max_num_of_mismatches, force_alignment,
 * normalize_reads, ref_prefix_len, ref_suffix_len.  One line per read (ReadLine above). */
int dvr_align_reads(const char* reference, const char* contig, int64_t ref_start, const char* const* haplotypes,
                    int n_haplotypes, const dvo_read* reads, int n_reads, const int32_t* options, double similarity,
                    char** out, uint64_t* out_len) {
  return Guard([&] {
    refdv::FastPassAligner aligner;
    refdv::AlignerOptions a;
    SetAlignerOptions(options, similarity, &a);
    aligner.set_options(a);
    aligner.set_normalize_reads(options[8] != 0);
    aligner.set_reference(reference);
    aligner.set_ref_start(contig, static_cast<uint64_t>(ref_start));
    aligner.set_ref_prefix_len(options[9]);
    aligner.set_ref_suffix_len(options[10]);
    aligner.set_haplotypes(Strings(haplotypes, n_haplotypes));
    std::vector<Read> protos(static_cast<size_t>(n_reads));
    for (int i = 0; i < n_reads; ++i) {
      MakeRead(reads[i], &protos[static_cast<size_t>(i)]);
      protos[static_cast<size_t>(i)].mutable_alignment()->mutable_position()->set_reference_name(contig);
    }
    auto realigned = aligner.AlignReads(protos);
    std::ostringstream text;
    for (const Read& r : *realigned) ReadLine(text, r);
    return TextOut(text, out, out_len);
  });
}

/* TrimReads (alt_aligned_pileup_lib.cc:231-248): one R line per kept read, followed by "P <original position>". */
int dvr_trim_reads(const dvo_read* reads, int n_reads, const char* contig, int64_t region_start, int64_t region_end,
                   int min_overlap, char** out, uint64_t* out_len) {
  return Guard([&] {
    std::vector<Read> protos(static_cast<size_t>(n_reads));
    std::vector<const Read*> ptrs;
    for (int i = 0; i < n_reads; ++i) {
      MakeRead(reads[i], &protos[static_cast<size_t>(i)]);
      protos[static_cast<size_t>(i)].mutable_alignment()->mutable_position()->set_reference_name(contig);
      ptrs.push_back(&protos[static_cast<size_t>(i)]);
    }
    nucleus::genomics::v1::Range region;
    region.set_reference_name(contig);
    region.set_start(region_start);
    region.set_end(region_end);
    std::vector<int64_t> original;
    const std::vector<Read> trimmed = refdv::TrimReads(ptrs, region, original, min_overlap);
    std::ostringstream text;
    for (size_t i = 0; i < trimmed.size(); ++i) {
      ReadLine(text, trimmed[i]);
      text << "P\t" << (i < original.size() ? original[i] : -1) << '\n';
    }
    return TextOut(text, out, out_len);
  });
}

/* RealignReadsToHaplotype (alt_aligned_pileup_lib.cc:278-313) over a reference handed in as one stretch, and
 * CalculateAlignmentRegion (:218-231) for a variant (start, reference_bases length) -> region_out[2]. */
int dvr_realign_reads_to_haplotype(const char* haplotype, const dvo_read* reads, int n_reads, const char* contig,
                                   int64_t ref_start, int64_t ref_end, int64_t contig_length, int64_t window_start,
                                   const char* window_bases, int64_t n_window_bases, const int32_t* options,
                                   double similarity, char** out, uint64_t* out_len) {
  return Guard([&] {
    WindowReference ref(contig, contig_length, window_start, std::string(window_bases, static_cast<size_t>(n_window_bases)));
    refdv::MakeExamplesOptions opts;
    SetAlignerOptions(options, similarity, opts.mutable_realigner_options()->mutable_aln_config());
    std::vector<Read> protos(static_cast<size_t>(n_reads));
    for (int i = 0; i < n_reads; ++i) {
      MakeRead(reads[i], &protos[static_cast<size_t>(i)]);
      protos[static_cast<size_t>(i)].mutable_alignment()->mutable_position()->set_reference_name(contig);
    }
    const std::vector<Read> realigned = refdv::RealignReadsToHaplotype(haplotype, protos, contig, ref_start, ref_end, ref, opts);
    std::ostringstream text;
    for (const Read& r : realigned) ReadLine(text, r);
    return TextOut(text, out, out_len);
  });
}

int dvr_calculate_alignment_region(const char* contig, int64_t contig_length, int64_t variant_start,
                                   int64_t n_reference_bases, int half_width, int64_t* region_out) {
  return Guard([&] {
    WindowReference ref(contig, contig_length, 0, "");
    nucleus::genomics::v1::Variant v;
    v.set_reference_name(contig);
    v.set_start(variant_start);
    v.set_end(variant_start + n_reference_bases);
    v.set_reference_bases(std::string(static_cast<size_t>(n_reference_bases), 'A'));
    const nucleus::genomics::v1::Range r = refdv::CalculateAlignmentRegion(v, half_width, ref);
    region_out[0] = r.start();
    region_out[1] = r.end();
    return 0;
  });
}

}  // extern "C"

extern "C" {

/* The stages of FastPassAligner::AlignReads up to (not including) RealignReadsToReference, then the state:
 *   H <haplotype index> <haplotype score> <ref_pos> <is_reference> <cigar>   per haplotype, in sorted order
 *   A <read> <position or -1> <score> <cigar>                               its read alignments */
int dvr_align_reads_state(const char* reference, const char* contig, int64_t ref_start, const char* const* haplotypes,
                          int n_haplotypes, const dvo_read* reads, int n_reads, const int32_t* options, double similarity,
                          char** out, uint64_t* out_len) {
  return Guard([&] {
    refdv::FastPassAligner aligner;
    refdv::AlignerOptions a;
    SetAlignerOptions(options, similarity, &a);
    aligner.set_options(a);
    aligner.set_normalize_reads(options[8] != 0);
    aligner.set_reference(reference);
    aligner.set_ref_start(contig, static_cast<uint64_t>(ref_start));
    aligner.set_ref_prefix_len(options[9]);
    aligner.set_ref_suffix_len(options[10]);
    aligner.set_haplotypes(Strings(haplotypes, n_haplotypes));
    std::vector<std::string> seqs;
    for (int i = 0; i < n_reads; ++i) seqs.emplace_back(reads[i].seq, static_cast<size_t>(reads[i].seq_len));
    aligner.set_reads(seqs);
    aligner.CalculateSswAlignmentScoreThreshold();
    aligner.BuildIndex();
    aligner.FastAlignReadsToHaplotypes();
    aligner.InitSswLib();
    aligner.AlignHaplotypesToReference();
    aligner.CalculatePositionMaps();
    aligner.SswAlignReadsToHaplotypes(static_cast<uint16_t>(aligner.get_ssw_alignment_score_threshold()));
    std::ostringstream text;
    text << "T\t" << aligner.get_ssw_alignment_score_threshold() << '\n';
    for (const auto& h : aligner.GetReadToHaplotypeAlignments()) {
      text << "H\t" << h.haplotype_index << '\t' << h.haplotype_score << '\t' << h.ref_pos << '\t' << (h.is_reference ? 1 : 0)
           << '\t' << h.cigar << '\n';
      for (size_t r = 0; r < h.read_alignment_scores.size(); ++r) {
        const auto& ra = h.read_alignment_scores[r];
        text << "A\t" << r << '\t' << (ra.position == refdv::ReadAlignment::kNotAligned ? -1 : static_cast<int>(ra.position))
             << '\t' << ra.score << '\t' << ra.cigar << '\n';
      }
    }
    return TextOut(text, out, out_len);
  });
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// The top of the hot path: ExamplesGenerator::WriteExamplesInRegion (deepvariant/make_examples_native.cc, compiled
// unmodified) -- InMemoryReader::Query per candidate, alt-allele combinations, one pileup per sample stacked,
// alt-aligned images, EncodeExample.  SURVEY.md 8(a) rows a1-a3, a12-a16.  Its tf.Examples are written through the
// ExampleWriter stand-in (kept in memory) and handed back as they are, length-prefixed.
// ---------------------------------------------------------------------------------------------------------------
#include "deepvariant/make_examples_native.h"

namespace {

std::vector<std::string> SplitTabs(const std::string& line) {
  std::vector<std::string> out;
  size_t i = 0;
  for (;;) {
    const size_t t = line.find('\t', i);
    out.push_back(line.substr(i, t == std::string::npos ? std::string::npos : t - i));
    if (t == std::string::npos) break;
    i = t + 1;
  }
  return out;
}

std::vector<std::string> SplitCommas(const std::string& s) {
  std::vector<std::string> out;
  if (s.empty()) return out;
  size_t i = 0;
  for (;;) {
    const size_t t = s.find(',', i);
    out.push_back(s.substr(i, t == std::string::npos ? std::string::npos : t - i));
    if (t == std::string::npos) break;
    i = t + 1;
  }
  return out;
}

}  // namespace

extern "C" {

/* `spec`: tab-separated lines.
 *   O <key> <value>        option: pic.<field> (PileupImageOptions), trim_reads_for_pileup, aln.<field> (AlignerOptions)
 *   M <role> <name> <pileup_height> <order,order,..> <alt_aligned_pileup> <blank enums,> <keep_only_window_spanning_reads>
 *                          one SampleOptions per line, in order
 *   C <start> <end> <reference_bases> <alt,alt,...> <call_set_name> <genotype,..>   one candidate
 *   S <allele> <read key,read key,...>                                              allele_support of the last C
 *   A <i,j;i;...>                                                                    make_examples_alt_allele_indices
 *   I <key> <int|float> <v,v,...>                                                    calls(0).info of the last C
 * reads: all samples' reads back to back, reads_per_sample[k] of them for sample k; sample_order and role as
 * WriteExamplesInRegion takes them.  out: per example a little-endian uint32 length + the serialized tf.Example;
 * image_shape_out[3]. */
int dvr_write_examples_in_region(const char* spec, const char* contig, int64_t contig_length, int64_t window_start,
                                 const char* window_bases, int64_t n_window_bases, const dvo_read* reads,
                                 const int32_t* reads_per_sample, int n_samples, const int32_t* sample_order,
                                 int n_sample_order, const char* role, const float* mean_coverage_per_sample,
                                 int32_t* image_shape_out, char** out, uint64_t* out_len) {
  return Guard([&] {
    refdv::MakeExamplesOptions options;
    auto* pic = options.mutable_pic_options();
    std::vector<refdv::DeepVariantCall> candidates;
    std::istringstream in(spec);
    std::string line;
    while (std::getline(in, line)) {
      if (line.empty()) continue;
      const std::vector<std::string> f = SplitTabs(line);
      if (f[0] == "O") {
        const std::string& k = f[1];
        const std::string& v = f[2];
        const int iv = std::atoi(v.c_str());
        const float fv = static_cast<float>(std::atof(v.c_str()));
        if (k == "pic.width") pic->set_width(iv);
        else if (k == "pic.height") pic->set_height(iv);
        else if (k == "pic.reference_band_height") pic->set_reference_band_height(iv);
        else if (k == "pic.base_color_offset_a_and_g") pic->set_base_color_offset_a_and_g(iv);
        else if (k == "pic.base_color_offset_t_and_c") pic->set_base_color_offset_t_and_c(iv);
        else if (k == "pic.base_color_stride") pic->set_base_color_stride(iv);
        else if (k == "pic.allele_supporting_read_alpha") pic->set_allele_supporting_read_alpha(fv);
        else if (k == "pic.allele_unsupporting_read_alpha") pic->set_allele_unsupporting_read_alpha(fv);
        else if (k == "pic.other_allele_supporting_read_alpha") pic->set_other_allele_supporting_read_alpha(fv);
        else if (k == "pic.reference_matching_read_alpha") pic->set_reference_matching_read_alpha(fv);
        else if (k == "pic.reference_mismatching_read_alpha") pic->set_reference_mismatching_read_alpha(fv);
        else if (k == "pic.indel_anchoring_base_char") pic->set_indel_anchoring_base_char(v);
        else if (k == "pic.reference_base_quality") pic->set_reference_base_quality(iv);
        else if (k == "pic.positive_strand_color") pic->set_positive_strand_color(iv);
        else if (k == "pic.negative_strand_color") pic->set_negative_strand_color(iv);
        else if (k == "pic.base_quality_cap") pic->set_base_quality_cap(iv);
        else if (k == "pic.mapping_quality_cap") pic->set_mapping_quality_cap(iv);
        else if (k == "pic.read_overlap_buffer_bp") pic->set_read_overlap_buffer_bp(iv);
        else if (k == "pic.min_base_quality") pic->mutable_read_requirements()->set_min_base_quality(iv);
        else if (k == "pic.min_mapping_quality") pic->mutable_read_requirements()->set_min_mapping_quality(iv);
        else if (k == "pic.multi_allelic_mode") pic->set_multi_allelic_mode(static_cast<refdv::PileupImageOptions::MultiAllelicMode>(iv));
        else if (k == "pic.random_seed") pic->set_random_seed(static_cast<uint32_t>(std::strtoul(v.c_str(), nullptr, 10)));
        else if (k == "pic.num_channels") pic->set_num_channels(iv);
        else if (k == "pic.sequencing_type") pic->set_sequencing_type(static_cast<refdv::PileupImageOptions::SequencingType>(iv));
        else if (k == "pic.alt_aligned_pileup") pic->set_alt_aligned_pileup(v);
        else if (k == "pic.types_to_alt_align") pic->set_types_to_alt_align(v);
        else if (k == "pic.sort_by_haplotypes") pic->set_sort_by_haplotypes(iv != 0);
        else if (k == "pic.hp_tag_for_assembly_polishing") pic->set_hp_tag_for_assembly_polishing(iv);
        else if (k == "pic.sort_by_alt_allele_support") pic->set_sort_by_alt_allele_support(iv != 0);
        else if (k == "pic.min_non_zero_allele_frequency") pic->set_min_non_zero_allele_frequency(fv);
        else if (k == "pic.channels") { for (const std::string& c : SplitCommas(v)) pic->add_channels(c); }
        else if (k == "trim_reads_for_pileup") options.set_trim_reads_for_pileup(iv != 0);
        else if (k.rfind("aln.", 0) == 0) {
          auto* a = options.mutable_realigner_options()->mutable_aln_config();
          if (k == "aln.match") a->set_match(iv);
          else if (k == "aln.mismatch") a->set_mismatch(iv);
          else if (k == "aln.gap_open") a->set_gap_open(iv);
          else if (k == "aln.gap_extend") a->set_gap_extend(iv);
          else if (k == "aln.kmer_size") a->set_kmer_size(iv);
          else if (k == "aln.max_num_of_mismatches") a->set_max_num_of_mismatches(iv);
          else if (k == "aln.realignment_similarity_threshold") a->set_realignment_similarity_threshold(std::atof(v.c_str()));
          else return fail("unknown option " + k);
        } else {
          return fail("unknown option " + k);
        }
      } else if (f[0] == "M") {
        auto* so = options.add_sample_options();
        so->set_role(f[1]);
        so->set_name(f[2]);
        so->set_pileup_height(std::atoi(f[3].c_str()));
        for (const std::string& x : SplitCommas(f[4])) so->add_order(std::atoi(x.c_str()));
        so->set_alt_aligned_pileup(f[5]);
        for (const std::string& x : SplitCommas(f[6])) so->add_channels_enum_to_blank(static_cast<refdv::DeepVariantChannelEnum>(std::atoi(x.c_str())));
        so->set_keep_only_window_spanning_reads(std::atoi(f[7].c_str()) != 0);
        if (f.size() > 9) {
          so->set_use_non_uniform_downsampling(std::atoi(f[8].c_str()) != 0);
          so->set_non_uniform_downsampling_threshold(std::atoi(f[9].c_str()));
        }
      } else if (f[0] == "C") {
        candidates.emplace_back();
        auto* v = candidates.back().mutable_variant();
        v->set_reference_name(contig);
        v->set_start(std::atoll(f[1].c_str()));
        v->set_end(std::atoll(f[2].c_str()));
        v->set_reference_bases(f[3]);
        for (const std::string& a : SplitCommas(f[4])) v->add_alternate_bases(a);
        if (!f[5].empty() || !f[6].empty()) {      // (a candidate without a VariantCall stays without one)
          auto* vc = v->add_calls();
          vc->set_call_set_name(f[5]);
          for (const std::string& g : SplitCommas(f[6])) vc->add_genotype(std::atoi(g.c_str()));
        }
      } else if (f[0] == "S") {
        auto& sr = (*candidates.back().mutable_allele_support())[f[1]];
        for (const std::string& name : SplitCommas(f[2])) sr.add_read_names(name);
      } else if (f[0] == "A") {
        size_t i = 0;
        const std::string& s = f[1];
        for (;;) {
          const size_t t = s.find(';', i);
          auto* idx = candidates.back().add_make_examples_alt_allele_indices();
          for (const std::string& x : SplitCommas(s.substr(i, t == std::string::npos ? std::string::npos : t - i))) idx->add_indices(std::atoi(x.c_str()));
          if (t == std::string::npos) break;
          i = t + 1;
        }
      } else if (f[0] == "I") {
        auto& lv = (*candidates.back().mutable_variant()->mutable_calls(0)->mutable_info())[f[1]];
        for (const std::string& x : SplitCommas(f[3])) {
          if (f[2] == "int") lv.add_values()->set_int_value(std::atoi(x.c_str()));
          else lv.add_values()->set_number_value(std::atof(x.c_str()));
        }
      } else {
        return fail("unknown line kind " + f[0]);
      }
    }
    options.set_mode(refdv::MakeExamplesOptions::CALLING);
    const std::string fasta = "dvref://reference", examples = "dvref://examples";
    options.set_reference_filename(fasta);
    options.set_examples_filename(examples);
    nucleus::IndexedFastaReader::Registered reg;
    reg.contig = contig;
    reg.contig_length = contig_length;
    reg.start = window_start;
    reg.bases.assign(window_bases, static_cast<size_t>(n_window_bases));
    nucleus::IndexedFastaReader::Registry()[fasta] = reg;
    nucleus::CapturedExamples::Store()[examples].clear();
    std::vector<int> image_shape;
    {
      refdv::ExamplesGenerator generator(options, {{role, examples}});
      std::vector<Read> protos;
      size_t total = 0;
      for (int s = 0; s < n_samples; ++s) total += static_cast<size_t>(reads_per_sample[s]);
      protos.resize(total);
      for (size_t i = 0; i < total; ++i) {
        MakeRead(reads[i], &protos[i]);
        protos[i].mutable_alignment()->mutable_position()->set_reference_name(contig);
      }
      std::vector<std::vector<nucleus::ConstProtoPtr<Read>>> per_sample(static_cast<size_t>(n_samples));
      size_t at = 0;
      for (int s = 0; s < n_samples; ++s) {
        for (int i = 0; i < reads_per_sample[s]; ++i) per_sample[static_cast<size_t>(s)].emplace_back(&protos[at++]);
      }
      std::vector<nucleus::ConstProtoPtr<refdv::DeepVariantCall>> cands;
      for (auto& c : candidates) cands.emplace_back(&c);
      const std::vector<int> order(sample_order, sample_order + n_sample_order);
      const std::vector<float> coverage(mean_coverage_per_sample, mean_coverage_per_sample + n_samples);
      generator.WriteExamplesInRegion(cands, per_sample, order, role, coverage, &image_shape);
    }
    for (size_t i = 0; i < 3; ++i) image_shape_out[i] = i < image_shape.size() ? image_shape[i] : 0;
    std::string blob;
    for (const std::string& ex : nucleus::CapturedExamples::Store()[examples]) {
      const uint32_t n = static_cast<uint32_t>(ex.size());
      blob.append(reinterpret_cast<const char*>(&n), 4);
      blob.append(ex);
    }
    *out = static_cast<char*>(std::malloc(blob.size() + 1));
    std::memcpy(*out, blob.data(), blob.size());
    *out_len = blob.size();
    return 0;
  });
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// The realigner's local assembly (deepvariant/realigner/debruijn_graph.cc) and the long-read chain's read phasing
// (deepvariant/direct_phasing.cc), compiled unmodified over a small Boost-Graph stand-in (shims/boost/graph/).
// SURVEY.md 8f row f4 and the phasing of section 13.
// ---------------------------------------------------------------------------------------------------------------
#include "deepvariant/direct_phasing.h"
#include "deepvariant/realigner/debruijn_graph.h"

extern "C" {

/* options: min_k, max_k, step_k, min_mapq, min_base_quality, min_edge_weight, max_num_paths, disable_graph_pruning.
 * Lines: "N" (no graph: no acyclic k), or "K <kmer size>", "H <haplotype>" (sorted), then "G" followed by the
 * graphviz dump on the remaining lines. */
int dvr_debruijn(const char* ref, const dvo_read* reads, int n_reads, const int32_t* options, char** out, uint64_t* out_len) {
  return Guard([&] {
    refdv::DeBruijnGraphOptions o;
    o.set_min_k(options[0]);
    o.set_max_k(options[1]);
    o.set_step_k(options[2]);
    o.set_min_mapq(options[3]);
    o.set_min_base_quality(options[4]);
    o.set_min_edge_weight(options[5]);
    o.set_max_num_paths(options[6]);
    o.set_disable_graph_pruning(options[7] != 0);
    std::vector<Read> protos(static_cast<size_t>(n_reads));
    std::vector<nucleus::ConstProtoPtr<const Read>> ptrs;
    for (int i = 0; i < n_reads; ++i) {
      MakeRead(reads[i], &protos[static_cast<size_t>(i)]);
      ptrs.emplace_back(&protos[static_cast<size_t>(i)]);
    }
    std::unique_ptr<refdv::DeBruijnGraph> graph = refdv::DeBruijnGraph::Build(ref, ptrs, o);
    std::ostringstream text;
    if (!graph) {
      text << "N\n";
    } else {
      text << "K\t" << graph->KmerSize() << '\n';
      for (const std::string& h : graph->CandidateHaplotypes()) text << "H\t" << h << '\n';
      text << "#GRAPHVIZ\n" << graph->GraphViz();
    }
    return TextOut(text, out, out_len);
  });
}

/* spec lines:  C <start> <end> <reference_bases> <alt,alt>   a candidate (strictly ordered by start)
 *              E <allele> <read name:is_low_quality,...>     its allele_support_ext
 *              F <read name:is_low_quality,...>              its ref_support_ext
 * phases_out[n_reads]; text: "V <position> <phase 1 bases> <phase 2 bases> <is_first_in_block>" per phased variant,
 * then "G" and the graphviz dump on the remaining lines. */
int dvr_phase_reads(const char* spec, const dvo_read* reads, int n_reads, int min_alleles_to_phase, int32_t* phases_out,
                    char** out, uint64_t* out_len) {
  return Guard([&] {
    std::vector<refdv::DeepVariantCall> candidates;
    std::istringstream in(spec);
    std::string line;
    auto fill = [](const std::string& list, refdv::DeepVariantCall::SupportingReadsExt* dst) {
      for (const std::string& item : SplitCommas(list)) {
        const size_t c = item.rfind(':');
        auto* info = dst->add_read_infos();
        info->set_read_name(item.substr(0, c));
        info->set_is_low_quality(item.substr(c + 1) == "1");
      }
    };
    while (std::getline(in, line)) {
      if (line.empty()) continue;
      const std::vector<std::string> f = SplitTabs(line);
      if (f[0] == "C") {
        candidates.emplace_back();
        auto* v = candidates.back().mutable_variant();
        v->set_reference_name("contig");
        v->set_start(std::atoll(f[1].c_str()));
        v->set_end(std::atoll(f[2].c_str()));
        v->set_reference_bases(f[3]);
        for (const std::string& a : SplitCommas(f[4])) v->add_alternate_bases(a);
      } else if (f[0] == "E") {
        fill(f.size() > 2 ? f[2] : "", &(*candidates.back().mutable_allele_support_ext())[f[1]]);
      } else if (f[0] == "F") {
        fill(f.size() > 1 ? f[1] : "", candidates.back().mutable_ref_support_ext());
      }
    }
    std::vector<Read> protos(static_cast<size_t>(n_reads));
    std::vector<nucleus::ConstProtoPtr<const Read>> ptrs;
    for (int i = 0; i < n_reads; ++i) {
      MakeRead(reads[i], &protos[static_cast<size_t>(i)]);
      ptrs.emplace_back(&protos[static_cast<size_t>(i)]);
    }
    refdv::DirectPhasingOptions o;
    o.set_min_alleles_to_phase(min_alleles_to_phase);
    refdv::DirectPhasing phasing(o);
    auto phases = phasing.PhaseReads(candidates, ptrs);
    if (!phases.ok()) return fail(std::string(phases.status().message()));
    const std::vector<int>& p = phases.ValueOrDie();
    for (int i = 0; i < n_reads; ++i) phases_out[i] = i < static_cast<int>(p.size()) ? p[static_cast<size_t>(i)] : 0;
    std::ostringstream text;
    for (const refdv::PhasedVariant& pv : phasing.GetPhasedVariants()) {
      text << "V\t" << pv.position << '\t' << pv.phase_1_bases << '\t' << pv.phase_2_bases << '\t' << (pv.is_first_in_block ? 1 : 0) << '\n';
    }
    text << "#GRAPHVIZ\n" << phasing.GraphViz();
    return TextOut(text, out, out_len);
  });
}

}  // extern "C"

extern "C" {

/* AlleleCounter::NormalizeAndAdd (allelecounter.cc:847-871) for every read, against a counter over [start, end):
 * per read one line "N <is_modified> <read_shift> <op:len,...>" (the normalised CIGAR, or the input one when it was
 * left alone). */
int dvr_normalize_cigars(const char* contig, int64_t contig_length, int64_t ref_start, const char* ref_bases,
                         int64_t n_ref_bases, int64_t start, int64_t end, const dvo_read* reads, int n_reads, char** out,
                         uint64_t* out_len) {
  return Guard([&] {
    WindowReference ref(contig, contig_length, ref_start, std::string(ref_bases, static_cast<size_t>(n_ref_bases)));
    refdv::AlleleCounterOptions co;
    co.set_partition_size(static_cast<int32_t>(end - start));
    co.set_normalize_reads(true);
    nucleus::genomics::v1::Range range;
    range.set_reference_name(contig);
    range.set_start(start);
    range.set_end(end);
    refdv::AlleleCounter counter(&ref, range, {}, co);
    std::ostringstream text;
    for (int i = 0; i < n_reads; ++i) {
      Read proto;
      MakeRead(reads[i], &proto);
      proto.mutable_alignment()->mutable_position()->set_reference_name(contig);
      std::unique_ptr<std::vector<CigarUnit>> norm(new std::vector<CigarUnit>());
      int shift = 0;
      counter.NormalizeAndAdd(proto, "s", norm, shift);
      const bool modified = !norm->empty();
      text << "N\t" << (modified ? 1 : 0) << '\t' << shift << '\t';
      if (modified) {
        for (size_t k = 0; k < norm->size(); ++k) text << (k ? "," : "") << static_cast<int>((*norm)[k].operation()) << ':' << (*norm)[k].operation_length();
      } else {
        for (int k = 0; k < proto.alignment().cigar_size(); ++k) {
          text << (k ? "," : "") << static_cast<int>(proto.alignment().cigar(k).operation()) << ':' << proto.alignment().cigar(k).operation_length();
        }
      }
      text << '\n';
    }
    return TextOut(text, out, out_len);
  });
}

}  // extern "C"
