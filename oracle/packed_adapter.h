// packed_adapter.h -- a packed batch item (dvo_packed_batch, the product's dv_batch layout) re-expanded into
// PROTO-SHAPED inputs: reads with synthesised names, a DeepVariantCall whose allele_support reproduces the packed
// support codes.  TEST INFRASTRUCTURE, shared by the oracle restatement (encoder_oracle.cpp) and by the wrapper
// around the reference's own encoder (ref_build/dvref_capi.cc), so that both are driven identically for full-size
// parity runs and for bench.py's cpu_baseline.
#ifndef DVO_PACKED_ADAPTER_H_
#define DVO_PACKED_ADAPTER_H_

#include <algorithm>
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
  }
  // A synthetic DeepVariantCall whose allele_support reproduces the packed codes (and, under
  // sort_by_alt_allele_support, the packed allele groups: alt number g of the variant is "ALT_G<g>", supported by the
  // reads of group g; the alt combination is the set of groups that hold a code-1 read).
  const bool grouped = b.list_group != nullptr && opt.sort_by_alt_allele_support;
  int n_groups = 0;
  if (grouped) {
    for (int i = 0; i < n; ++i) {
      if (b.list_code[l0 + i] != 0) n_groups = std::max(n_groups, static_cast<int>(b.list_group[l0 + i]) + 1);
    }
  }
  std::vector<std::string> alt_names;
  std::vector<std::vector<std::string>> alt_support;
  std::vector<char> in_combo;
  if (grouped && n_groups > 0) {
    alt_names.resize(static_cast<size_t>(n_groups));
    alt_support.resize(static_cast<size_t>(n_groups));
    in_combo.assign(static_cast<size_t>(n_groups), 0);
    for (int g = 0; g < n_groups; ++g) alt_names[static_cast<size_t>(g)] = "ALT_G" + std::to_string(g);
    // A group is in the combination when its reads all carry code 1 (a read listed under an alt of the combination
    // supports it: code 1; a group outside it holds code-2 reads -- and possibly reads that are ALSO listed under
    // an alt of the combination: code 1 with the group of their last listing, pileup_image_native.cc:346-361).
    std::vector<char> has1(static_cast<size_t>(n_groups), 0), has2(static_cast<size_t>(n_groups), 0);
    for (int i = 0; i < n; ++i) {
      const int code = b.list_code[l0 + i];
      if (code == 1) has1[b.list_group[l0 + i]] = 1;
      if (code == 2) has2[b.list_group[l0 + i]] = 1;
    }
    for (int g = 0; g < n_groups; ++g) in_combo[static_cast<size_t>(g)] = has1[static_cast<size_t>(g)] && !has2[static_cast<size_t>(g)];
    // Reads that carry code 1 inside a group outside the combination are ALSO listed under an alt of the
    // combination (their group is that of their last listing).  They join a synthetic alt of the combination placed
    // in FRONT of all groups: every real group moves up by one, so the groups' order -- all that row sorting uses --
    // is unchanged, and "last listing wins" still gives such a read its own group.
    std::vector<std::string> doubly;
    for (int i = 0; i < n; ++i) {
      const int code = b.list_code[l0 + i];
      if (code == 0) continue;
      const int g = b.list_group[l0 + i];
      alt_support[static_cast<size_t>(g)].push_back(names[i] + "/0");
      if (code == 1 && !in_combo[static_cast<size_t>(g)]) doubly.push_back(names[i] + "/0");
    }
    if (!doubly.empty()) {
      alt_names.insert(alt_names.begin(), "ALT_OF_THE_COMBINATION");
      alt_support.insert(alt_support.begin(), doubly);
      in_combo.insert(in_combo.begin(), 1);
    }
  } else {
    alt_names = {"ALT_IN", "ALT_OTHER"};
    alt_support.resize(2);
    in_combo = {1, 0};
    for (int i = 0; i < n; ++i) {
      if (b.list_code[l0 + i] == 1) alt_support[0].push_back(names[i] + "/0");
      if (b.list_code[l0 + i] == 2) alt_support[1].push_back(names[i] + "/0");
    }
  }
  std::vector<const char*> alts, support_names, combo;
  std::vector<int32_t> support_offsets{0};
  for (size_t g = 0; g < alt_names.size(); ++g) {
    alts.push_back(alt_names[g].c_str());
    for (const std::string& s : alt_support[g]) support_names.push_back(s.c_str());
    support_offsets.push_back(static_cast<int32_t>(support_names.size()));
    if (in_combo[g]) combo.push_back(alt_names[g].c_str());
  }
  if (combo.empty()) combo.push_back("ALT_NOT_IN_THE_VARIANT");
  dvo_call call;
  memset(&call, 0, sizeof(call));
  call.variant_start = b.item_variant_start[item];
  call.n_alts = static_cast<int32_t>(alts.size());
  call.alts = alts.data();
  call.n_support = static_cast<int32_t>(alts.size());
  call.support_alleles = alts.data();
  call.support_offsets = support_offsets.data();
  call.support_names = support_names.data();

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
  return build(call, ref, reads.data(), n, b.item_image_start[item], combo.data(), static_cast<int>(combo.size()), h, mean_cov,
               sort_pos.data(), blank.data(), static_cast<int>(blank.size()));
}

}  // namespace dvo_adapter

#endif  // DVO_PACKED_ADAPTER_H_
