// read_table.h -- the packed read table the alignment-file decoders (bam_reader.cpp, cram_reader.cpp)
// fill: dv_batch's structure-of-arrays read layout on the host (include/dvhip.h, dv_read_table).
#ifndef DV_READ_TABLE_H_
#define DV_READ_TABLE_H_

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

struct dv_read_table {
  std::vector<int32_t> pos, frag_len, hp;
  std::vector<uint32_t> seq_off, cigar_off, cigar, name_rank, name_off;
  std::vector<uint8_t> mapq, flags, read_number, bases, quals;
  std::vector<int64_t> end;
  std::vector<char> names;  // NUL-terminated, concatenated
  // Optional per-base planes, parallel to `bases`, derived from aux tags (aux_planes.h) when the read requirements
  // ask for them: base modifications (MM / ML / MN; flags carry DV_READ_HAS_5MC / _6MA per read) and the Ultima
  // flow-space tags (tp values; t0 characters - 33; flow_present: bit 0 = the read has tp, bit 1 = it has t0).
  bool with_mods = false, with_flow = false;
  std::vector<uint8_t> mod_5mc, mod_6ma, t0, flow_present;
  std::vector<int8_t> tp;
};

namespace dv {

// Dense rank of every read under the reference's tuple<string, int> ordering
// (fragment_name, read_number): what SortImageRows breaks position ties with.
inline void rank_read_names(dv_read_table* t) {
  const size_t n = t->pos.size();
  std::vector<uint32_t> order(n);
  std::iota(order.begin(), order.end(), 0u);
  auto key_less = [&](uint32_t a, uint32_t b) {
    const int c = std::strcmp(&t->names[t->name_off[a]], &t->names[t->name_off[b]]);
    return c != 0 ? c < 0 : t->read_number[a] < t->read_number[b];
  };
  std::sort(order.begin(), order.end(), key_less);
  t->name_rank.assign(n, 0);
  uint32_t rank = 0;
  for (size_t i = 0; i < n; ++i) {
    if (i && key_less(order[i - 1], order[i])) ++rank;
    t->name_rank[order[i]] = rank;
  }
}

}  // namespace dv

#endif  // DV_READ_TABLE_H_
