// C ABI of the read realigner (include/dvhip.h, "realigner" section): dv::FastPassAligner
// behind an opaque handle, CIGARs as text for the stage-level entry points the tests use and
// as (length << 4 | op) words -- dv_batch's CIGAR encoding -- for the product entry point.
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "debruijn_graph.h"
#include "direct_phasing.h"
#include "dv_internal.h"
#include "fast_pass_aligner.h"

struct dv_aligner {
  dv::FastPassAligner a;
  std::vector<uint32_t> cigar_words;
};

struct dv_debruijn_graph {
  std::unique_ptr<dv::DeBruijnGraph> g;
  std::vector<std::string> haplotypes;
  std::vector<const char*> haplotype_ptrs;
  std::string dot;
};

namespace {

int copy_text(const std::string& s, char* out, int32_t cap) {
  if (!out || cap <= 0 || static_cast<size_t>(cap) <= s.size()) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "text buffer too small");
  }
  std::memcpy(out, s.c_str(), s.size() + 1);
  return DV_OK;
}

int fill(const dv::ReadAlignment& ra, dv_read_alignment* out) {
  out->position = ra.position == dv::ReadAlignment::kNotAligned ? -1 : ra.position;
  out->score = ra.score;
  return copy_text(ra.cigar, out->cigar, sizeof(out->cigar));   // never truncated silently
}

std::vector<std::string> strings(int32_t n, const char* const* v) {
  std::vector<std::string> out;
  out.reserve(n);
  for (int32_t i = 0; i < n; ++i) out.emplace_back(v[i] ? v[i] : "");
  return out;
}

}  // namespace

extern "C" {

int dv_aligner_create(const dv_aligner_options* o, dv_aligner** out) {
  if (!out) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_aligner_create: null");
  auto h = std::make_unique<dv_aligner>();
  if (o) {
    dv::AlignerOptions ao;
    ao.match = o->match;
    ao.mismatch = o->mismatch;
    ao.gap_open = o->gap_open;
    ao.gap_extend = o->gap_extend;
    ao.kmer_size = o->kmer_size;
    ao.read_size = o->read_size;
    ao.max_num_of_mismatches = o->max_num_of_mismatches;
    ao.similarity_threshold = o->realignment_similarity_threshold;
    ao.force_alignment = o->force_alignment != 0;
    std::string error;
    if (!h->a.set_options(ao, &error)) return dv::fail(DV_ERR_INVALID_ARGUMENT, error);
    h->a.set_normalize_reads(o->normalize_reads != 0);
    h->a.set_ref_prefix_len(o->ref_prefix_len);
    h->a.set_ref_suffix_len(o->ref_suffix_len);
  }
  *out = h.release();
  return DV_OK;
}

void dv_aligner_destroy(dv_aligner* h) { delete h; }

int dv_aligner_set_reference(dv_aligner* h, const char* reference, int64_t ref_start) {
  if (!h || !reference || ref_start < 0) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_aligner_set_reference");
  h->a.set_reference(reference);
  h->a.set_ref_start(static_cast<uint64_t>(ref_start));
  return DV_OK;
}

int dv_aligner_set_haplotypes(dv_aligner* h, int32_t n, const char* const* haplotypes) {
  if (!h || n < 0 || (n && !haplotypes)) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_aligner_set_haplotypes");
  for (int32_t i = 0; i < n; ++i) {
    if (haplotypes[i] && std::strlen(haplotypes[i]) >= 0xffff) {
      return dv::fail(DV_ERR_INVALID_ARGUMENT, "haplotypes are limited to 65534 bases (16-bit read offsets)");
    }
  }
  h->a.set_haplotypes(strings(n, haplotypes));
  return DV_OK;
}

int dv_aligner_set_reads(dv_aligner* h, int32_t n, const char* const* reads) {
  if (!h || n < 0 || (n && !reads)) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_aligner_set_reads");
  h->a.set_reads(strings(n, reads));
  return DV_OK;
}

int dv_aligner_align_reads(dv_aligner* h, int32_t n, const char* const* sequences, dv_realigned_read* out,
                           const uint32_t** cigar) {
  if (!h || n < 0 || (n && (!sequences || !out)) || !cigar) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_aligner_align_reads");
  }
  const std::vector<dv::RealignedRead> res = h->a.align_reads(strings(n, sequences));
  h->cigar_words.clear();
  for (int32_t i = 0; i < n; ++i) {
    out[i].status = res[i].status;
    out[i].position = res[i].position;
    out[i].cigar_off = static_cast<uint32_t>(h->cigar_words.size());
    out[i].n_cigar = static_cast<int32_t>(res[i].cigar.size());
    for (const dv::CigarOp& op : res[i].cigar) {
      h->cigar_words.push_back((static_cast<uint32_t>(op.length) << 4) | static_cast<uint32_t>(op.op));
    }
  }
  *cigar = h->cigar_words.data();
  return DV_OK;
}

int dv_aligner_stage(dv_aligner* h, int32_t stage, int32_t arg) {
  if (!h) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_aligner_stage: null");
  switch (stage) {
    case DV_ALIGNER_BUILD_INDEX: h->a.build_index(); break;
    case DV_ALIGNER_INIT_LOCAL_ALIGNER: h->a.init_local_aligner(); break;
    case DV_ALIGNER_ALIGN_HAPLOTYPES: h->a.align_haplotypes_to_reference(); break;
    case DV_ALIGNER_POSITION_MAPS: h->a.calculate_position_maps(); break;
    case DV_ALIGNER_LOCAL_ALIGN_READS: h->a.local_align_reads_to_haplotypes(arg); break;
    case DV_ALIGNER_SCORE_THRESHOLD: h->a.calculate_score_threshold(); break;
    default: return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_aligner_stage: unknown stage");
  }
  return DV_OK;
}

int dv_aligner_fast_align(dv_aligner* h, const char* haplotype, int32_t* haplotype_score,
                          dv_read_alignment* out) {
  if (!h || !haplotype || !haplotype_score) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_aligner_fast_align");
  std::vector<dv::ReadAlignment> ra(h->a.reads().size());
  int score = *haplotype_score;
  h->a.fast_align_reads_to_haplotype(haplotype, &score, &ra);
  *haplotype_score = score;
  for (size_t i = 0; out && i < ra.size(); ++i) {
    if (int rc = fill(ra[i], &out[i])) return rc;
  }
  return DV_OK;
}

int dv_aligner_haplotype_info(const dv_aligner* h, int32_t k, int32_t* haplotype_index, int32_t* haplotype_score,
                              int64_t* ref_pos, int32_t* is_reference, char* cigar, int32_t cigar_cap) {
  if (!h || k < 0 || static_cast<size_t>(k) >= h->a.haplotype_alignments().size()) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_aligner_haplotype_info: index");
  }
  const dv::HaplotypeAlignment& ha = h->a.haplotype_alignments()[k];
  if (haplotype_index) *haplotype_index = static_cast<int32_t>(ha.haplotype_index);
  if (haplotype_score) *haplotype_score = ha.haplotype_score;
  if (ref_pos) *ref_pos = static_cast<int64_t>(ha.ref_pos);
  if (is_reference) *is_reference = ha.is_reference ? 1 : 0;
  return cigar ? copy_text(ha.cigar, cigar, cigar_cap) : DV_OK;
}

int dv_aligner_read_alignment(const dv_aligner* h, int32_t k, int32_t read, dv_read_alignment* out) {
  if (!h || !out || k < 0 || static_cast<size_t>(k) >= h->a.haplotype_alignments().size() || read < 0 ||
      static_cast<size_t>(read) >= h->a.haplotype_alignments()[k].reads.size()) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_aligner_read_alignment: index");
  }
  return fill(h->a.haplotype_alignments()[k].reads[read], out);
}

int dv_aligner_merge_alignment(const dv_aligner* h, int32_t read, int32_t position, const char* read_cigar,
                               const char* haplotype_cigar, char* out, int32_t cap) {
  if (!h || !read_cigar || !haplotype_cigar || position < 0 || position >= 0xffff) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_aligner_merge_alignment");
  }
  dv::ReadAlignment ra;
  ra.position = static_cast<uint16_t>(position);
  ra.cigar = read_cigar;
  dv::Cigar merged;
  std::string error;
  if (!h->a.calculate_read_to_ref_alignment(static_cast<size_t>(read), ra, dv::parse_cigar(haplotype_cigar),
                                            &merged, &error)) {
    return dv::fail(DV_ERR_BAD_INPUT, error);
  }
  return copy_text(dv::cigar_text(merged), out, cap);
}

int dv_aligner_is_normalized(const dv_aligner* h, const char* cigar, int32_t ref_offset, const char* read) {
  if (!h || !cigar || !read) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_aligner_is_normalized");
  return h->a.is_alignment_normalized(dv::parse_cigar(cigar), ref_offset, read) ? 1 : 0;
}

int dv_aligner_score_threshold(const dv_aligner* h) { return h ? h->a.score_threshold() : -1; }

int dv_aligner_kmer_occurrences(const dv_aligner* h, const char* kmer, int32_t cap, int32_t* reads,
                                int32_t* offsets) {
  if (!h || !kmer) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_aligner_kmer_occurrences");
  if (*kmer == '\0') return static_cast<int>(h->a.index_size());
  const auto occ = h->a.kmer_occurrences(kmer);
  for (size_t i = 0; i < occ.size() && static_cast<int32_t>(i) < cap; ++i) {
    if (reads) reads[i] = static_cast<int32_t>(occ[i].first);
    if (offsets) offsets[i] = static_cast<int32_t>(occ[i].second);
  }
  return static_cast<int>(occ.size());
}

int dv_positions_map(const char* cigar, int32_t haplotype_size, int32_t* out) {
  if (!cigar || haplotype_size < 0 || !out) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_positions_map");
  const std::vector<int> m = dv::positions_map(cigar, static_cast<size_t>(haplotype_size));
  for (size_t i = 0; i < m.size(); ++i) out[i] = m[i];
  return DV_OK;
}

int dv_merge_cigar_op(char* cigar, int32_t cap, char op, int32_t length, int32_t read_len) {
  if (!cigar) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_merge_cigar_op");
  dv::Cigar c = dv::parse_cigar(cigar);
  const int kind = op == 'M' || op == '=' || op == 'X' ? dv::kOpMatch
                   : op == 'I' ? dv::kOpInsert : op == 'D' ? dv::kOpDelete : op == 'S' ? dv::kOpSoftClip : -1;
  if (kind < 0) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_merge_cigar_op: op");
  dv::merge_cigar_op({kind, length}, read_len, &c);
  return copy_text(dv::cigar_text(c), cigar, cap);
}

int dv_local_align(const char* reference, const char* query, int32_t match, int32_t mismatch, int32_t gap_open,
                   int32_t gap_extend, dv_local_alignment* out) {
  if (!reference || !query || !out) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_local_align");
  dv::LocalAligner a(match, mismatch, gap_open, gap_extend);
  a.set_reference(reference);
  dv::LocalAlignment r;
  if (!a.align(query, &r)) return dv::fail(DV_ERR_BAD_INPUT, "dv_local_align: empty or oversized sequences");
  out->score = r.score;
  out->ref_begin = r.ref_begin;
  out->ref_end = r.ref_end;
  out->query_begin = r.query_begin;
  out->query_end = r.query_end;
  out->mismatches = r.mismatches;
  return copy_text(r.cigar, out->cigar, sizeof(out->cigar));
}

// n queries against one reference through the 16-lane batch path; out[k].score < 0 marks a
// query the aligner refuses (empty, or an oversized sub-problem)
int dv_local_align_many(const char* reference, int32_t n, const char* const* queries, int32_t match,
                        int32_t mismatch, int32_t gap_open, int32_t gap_extend, dv_local_alignment* out) {
  if (!reference || n < 0 || (n > 0 && (!queries || !out))) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_local_align_many");
  dv::LocalAligner a(match, mismatch, gap_open, gap_extend);
  a.set_reference(reference);
  std::vector<dv::LocalAlignment> results;
  std::vector<char> ok;
  a.align_many_to_reference(strings(n, queries), &results, &ok);
  for (int32_t k = 0; k < n; ++k) {
    const dv::LocalAlignment& r = results[k];
    out[k].score = ok[k] ? r.score : -1;
    out[k].ref_begin = r.ref_begin;
    out[k].ref_end = r.ref_end;
    out[k].query_begin = r.query_begin;
    out[k].query_end = r.query_end;
    out[k].mismatches = r.mismatches;
    out[k].cigar[0] = '\0';
    if (ok[k]) {
      const int rc = copy_text(r.cigar, out[k].cigar, sizeof(out[k].cigar));
      if (rc != DV_OK) return rc;
    }
  }
  return DV_OK;
}

// ---- local assembly (debruijn_graph.cpp)

int dv_debruijn_build(const char* ref, int64_t ref_len, const uint8_t* bases, const uint8_t* quals, int64_t n_bases,
                      const uint32_t* read_seq_off, const uint8_t* read_mapq, int32_t n_table_reads,
                      const int32_t* reads, int32_t n_reads, const dv_debruijn_options* o,
                      dv_debruijn_graph** out) {
  if (!out) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_debruijn_build: null out");
  *out = nullptr;
  if (!ref || ref_len < 0 || !o || n_reads < 0 || n_table_reads < 0 || n_bases < 0 ||
      (n_reads > 0 && (!bases || !quals || !read_seq_off || !read_mapq || !reads))) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_debruijn_build: null argument");
  }
  if (o->step_k <= 0 || o->min_k <= 0) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_debruijn_build: min_k and step_k must be positive");
  }
  std::vector<dv::AssemblyRead> rs;
  rs.reserve(n_reads);
  for (int32_t i = 0; i < n_reads; ++i) {
    const int32_t r = reads[i];
    if (r < 0 || r >= n_table_reads || read_seq_off[r + 1] < read_seq_off[r] || read_seq_off[r + 1] > n_bases) {
      return dv::fail(DV_ERR_BAD_INPUT, "dv_debruijn_build: read index or its bases outside the table");
    }
    const uint32_t a = read_seq_off[r], b = read_seq_off[r + 1];
    rs.push_back(dv::AssemblyRead{std::string_view(reinterpret_cast<const char*>(bases) + a, b - a), quals + a,
                                  read_mapq[r]});
  }
  dv::DeBruijnOptions opt;
  opt.min_k = o->min_k;
  opt.max_k = o->max_k;
  opt.step_k = o->step_k;
  opt.min_mapq = o->min_mapq;
  opt.min_base_quality = o->min_base_quality;
  opt.min_edge_weight = o->min_edge_weight;
  opt.max_num_paths = o->max_num_paths;
  opt.disable_graph_pruning = o->disable_graph_pruning != 0;
  auto g = dv::DeBruijnGraph::build(std::string_view(ref, static_cast<size_t>(ref_len)), rs, opt);
  if (!g) return DV_OK;            // no acyclic graph for any k: *out stays NULL
  auto h = std::make_unique<dv_debruijn_graph>();
  h->g = std::move(g);
  *out = h.release();
  return DV_OK;
}

void dv_debruijn_destroy(dv_debruijn_graph* h) { delete h; }

int dv_debruijn_kmer_size(const dv_debruijn_graph* h) {
  if (!h) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_debruijn_kmer_size: null");
  return h->g->kmer_size();
}

int dv_debruijn_haplotypes(dv_debruijn_graph* h, int32_t* n, const char* const** haplotypes) {
  if (!h || !n || !haplotypes) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_debruijn_haplotypes: null");
  h->haplotypes = h->g->candidate_haplotypes();
  h->haplotype_ptrs.clear();
  for (const std::string& s : h->haplotypes) h->haplotype_ptrs.push_back(s.c_str());
  *n = static_cast<int32_t>(h->haplotypes.size());
  *haplotypes = h->haplotype_ptrs.data();
  return DV_OK;
}

int dv_debruijn_graphviz(dv_debruijn_graph* h, const char** text) {
  if (!h || !text) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_debruijn_graphviz: null");
  h->dot = h->g->graphviz();
  *text = h->dot.c_str();
  return DV_OK;
}

// ---- read phasing (direct_phasing.cpp)

int dv_phase_reads(const dv_phasing_candidate* candidates, int32_t n_candidates, const dv_phasing_allele* alleles,
                   int32_t n_alleles, const char* bases, int64_t n_bases, const int32_t* support_reads,
                   const uint8_t* support_low_quality, int64_t n_support, int32_t n_reads,
                   int32_t min_alleles_to_phase, int32_t* read_phases, int32_t* allele_phases,
                   uint8_t* allele_flags, char* graphviz, int32_t graphviz_cap) {
  if (n_candidates < 0 || n_alleles < 0 || n_reads < 0 || n_bases < 0 || n_support < 0 ||
      (n_candidates > 0 && !candidates) || (n_alleles > 0 && !alleles) || (n_reads > 0 && !read_phases) ||
      (n_support > 0 && (!support_reads || !support_low_quality)) || (n_bases > 0 && !bases)) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_phase_reads: null or negative argument");
  }
  std::vector<dv::PhasingCandidate> cs(n_candidates);
  for (int32_t i = 0; i < n_candidates; ++i) {
    const dv_phasing_candidate& c = candidates[i];
    if (c.allele_off < 0 || c.n_alleles < 0 || static_cast<int64_t>(c.allele_off) + c.n_alleles > n_alleles) {
      return dv::fail(DV_ERR_BAD_INPUT, "dv_phase_reads: candidate allele range outside the allele table");
    }
    cs[i].start = c.start;
    cs[i].end = c.end;
    for (int32_t k = 0; k < c.n_alleles; ++k) {
      const dv_phasing_allele& a = alleles[c.allele_off + k];
      if (a.bases_off < 0 || a.bases_len < 0 || a.bases_off + a.bases_len > n_bases || a.support_off < 0 ||
          a.n_support < 0 || a.support_off + a.n_support > n_support) {
        return dv::fail(DV_ERR_BAD_INPUT, "dv_phase_reads: allele range outside its table");
      }
      dv::PhasingAllele pa;
      pa.bases.assign(bases + a.bases_off, static_cast<size_t>(a.bases_len));
      pa.is_ref = a.is_ref != 0;
      for (int64_t r = a.support_off; r < a.support_off + a.n_support; ++r) {
        if (support_reads[r] >= n_reads) return dv::fail(DV_ERR_BAD_INPUT, "dv_phase_reads: read index out of range");
        pa.support.push_back(dv::PhasingReadSupport{support_reads[r], support_low_quality[r] != 0});
      }
      cs[i].alleles.push_back(std::move(pa));
    }
  }
  dv::DirectPhasing phasing(min_alleles_to_phase);
  std::vector<int> phases;
  std::string error;
  if (!phasing.phase_reads(cs, n_reads, &phases, &error)) return dv::fail(DV_ERR_BAD_INPUT, error);
  for (int32_t r = 0; r < n_reads; ++r) read_phases[r] = phases[r];
  if (allele_phases || allele_flags) {
    for (int32_t i = 0; i < n_candidates; ++i) {
      const auto& per_allele = phasing.allele_phases()[i];
      for (size_t k = 0; k < per_allele.size(); ++k) {
        const int32_t at = candidates[i].allele_off + static_cast<int32_t>(k);
        if (allele_phases) allele_phases[at] = per_allele[k].in_graph ? per_allele[k].phase : -1;
        if (allele_flags) allele_flags[at] = per_allele[k].is_first_in_block ? 1 : 0;
      }
    }
  }
  if (graphviz) return copy_text(phasing.graphviz(), graphviz, graphviz_cap);
  return DV_OK;
}

}  // extern "C"
