// packed_adapter.h -- a packed batch item (dvo_packed_batch, the product's dv_batch layout) re-expanded into
// PROTO-SHAPED inputs: reads with synthesised names, a DeepVariantCall whose allele_support reproduces the packed
// support codes.  TEST INFRASTRUCTURE, shared by the oracle restatement (encoder_oracle.cpp) and by the wrapper
// around the reference's own encoder (ref_build/dvref_capi.cc), so that both are driven identically for full-size
// parity runs and for bench.py's cpu_baseline.
#ifndef DVO_PACKED_ADAPTER_H_
#define DVO_PACKED_ADAPTER_H_

#include <climits>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "dvo.h"

namespace dvo_adapter {

// build(call, ref_bases, reads, n_reads, image_start_pos, alt_alleles, n_alt_alleles, pileup_height, mean_coverage,
//       alignment_positions, channels_to_blank, n_blank) -> rows kept, < 0 on error.
template <class Build>
int ExpandPackedItem(const dvo_options& opt, const dvo_packed_batch& b, int item, std::string* error, Build build) {
  const uint32_t l0 = b.item_list_off[item], l1 = b.item_list_off[item + 1];
  const int n = static_cast<int>(l1 - l0);
  std::vector<dvo_read> reads(n);
  std::vector<std::string> names(n);
  std::vector<std::vector<int32_t>> ops(n);
  std::vector<std::vector<int64_t>> lens(n);
  std::vector<int64_t> sort_pos(n);
  // A synthetic DeepVariantCall whose allele_support reproduces the codes:
  // "ALT_IN" is in alt_alleles (code 1), "ALT_OTHER" is not (code 2).
  std::vector<std::string> in_names, other_names;
  for (int i = 0; i < n; ++i) {
    const uint32_t r = b.list_read[l0 + i];
    char buf[16];
    snprintf(buf, sizeof(buf), "%010u", b.read_name_rank[r]);
    names[i] = buf;
    dvo_read& rd = reads[i];
    memset(&rd, 0, sizeof(rd));
    rd.fragment_name = names[i].c_str();
    rd.read_number = 0;
    rd.position = b.read_pos[r];
    sort_pos[i] = b.read_sort_pos ? b.read_sort_pos[r] : b.read_pos[r];
    rd.mapping_quality = b.read_mapq[r];
    rd.reverse_strand = b.read_flags[r] & 1;
    rd.supplementary = (b.read_flags[r] >> 1) & 1;
    rd.fragment_length = b.read_frag_len[r];
    const uint32_t s0 = b.read_seq_off[r], s1 = b.read_seq_off[r + 1];
    rd.seq = reinterpret_cast<const char*>(b.bases + s0);
    rd.seq_len = s1 - s0;
    rd.qual = b.quals + s0;
    rd.qual_len = s1 - s0;
    for (uint32_t c = b.read_cigar_off[r]; c < b.read_cigar_off[r + 1]; ++c) {
      ops[i].push_back(b.cigar[c] & 0xF);
      lens[i].push_back(b.cigar[c] >> 4);
    }
    rd.cigar_ops = ops[i].data();
    rd.cigar_lens = lens[i].data();
    rd.n_cigar = ops[i].size();
    if (b.read_hp[r] != INT32_MIN) {
      rd.hp_present = 1;
      rd.hp_n_values = 1;
      rd.hp_is_int = 1;
      rd.hp_value = b.read_hp[r];
    }
    if (b.mod_5mc && (b.read_flags[r] & 4)) {
      rd.mod_5mc = b.mod_5mc + s0;
      rd.mod_5mc_len = s1 - s0;
    }
    if (b.mod_6ma && (b.read_flags[r] & 8)) {
      rd.mod_6ma = b.mod_6ma + s0;
      rd.mod_6ma_len = s1 - s0;
    }
    const std::string key = names[i] + "/0";
    if (b.list_code[l0 + i] == 1) in_names.push_back(key);
    if (b.list_code[l0 + i] == 2) other_names.push_back(key);
  }
  if (b.list_group != nullptr && opt.sort_by_alt_allele_support) {
    *error = "packed adapter: sort_by_alt_allele_support not supported";
    return -1;
  }
  const char* alts[2] = {"ALT_IN", "ALT_OTHER"};
  std::vector<const char*> support_names;
  for (auto& s : in_names) support_names.push_back(s.c_str());
  for (auto& s : other_names) support_names.push_back(s.c_str());
  int32_t support_offsets[3] = {0, static_cast<int32_t>(in_names.size()),
                                static_cast<int32_t>(support_names.size())};
  dvo_call call;
  memset(&call, 0, sizeof(call));
  call.variant_start = b.item_variant_start[item];
  call.n_alts = 2;
  call.alts = alts;
  call.n_support = 2;
  call.support_alleles = alts;
  call.support_offsets = support_offsets;
  call.support_names = support_names.data();
  const char* alt_alleles[1] = {"ALT_IN"};

  const int h = b.item_height[item];
  std::string ref(reinterpret_cast<const char*>(b.ref_windows) +
                      static_cast<size_t>(b.item_ref_idx[item]) * opt.width,
                  opt.width);
  std::vector<int32_t> blank;
  if (b.item_blank_mask != nullptr) {
    for (int c = 0; c < opt.n_channels; ++c) {
      if ((b.item_blank_mask[item] >> c) & 1u) blank.push_back(opt.channels[c]);
    }
  }
  const float mean_cov = b.item_mean_coverage ? b.item_mean_coverage[item] : 0.0f;
  return build(call, ref, reads.data(), n, b.item_image_start[item], alt_alleles, 1, h, mean_cov, sort_pos.data(),
               blank.data(), static_cast<int>(blank.size()));
}

}  // namespace dvo_adapter

#endif  // DVO_PACKED_ADAPTER_H_
