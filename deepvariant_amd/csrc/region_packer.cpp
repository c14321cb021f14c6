// region_packer.cpp -- candidates + the reads of one region -> the item / list arrays of a
// dv_batch, on the host, without per-read or per-candidate Python.
//
// Replaces what ExamplesGenerator::CreateAndWriteExamplesForCandidate decides BEFORE pixels
// are drawn (deepvariant/make_examples_native.cc:632-736):
//   * the reads of each candidate: InMemoryReader::Query + nucleus::ReadOverlapsRegion on
//     [variant.start - read_overlap_buffer_bp, variant.end + read_overlap_buffer_bp)
//     (:645-648,802-810; third_party/nucleus/util/utils.cc:172-188), in caller order;
//   * one item per alt-allele combination (AltAlleleCombinations, :191-267 -- the
//     combinations themselves are passed in, as bit masks over alternate_bases);
//   * per (item, read) the ReadSupportsAlt code (channels/read_supports_variant_channel.cc:
//     54-116): 0 = none, 1 = supports an alt of this combination, 2 = supports another alt,
//     where a read listed under several alts counts for the FIRST alt in
//     variant.alternate_bases order; and, for sort_by_alt_allele_support, the allele group
//     (pileup_image_native.cc:346-393: the LAST listing alt wins, default = number of alts).
// Read keys are "<fragment_name>/<read_number>" (utils.cc ReadKey), matched through a hash of
// the key bytes with a full comparison on every hit -- no string ever reaches the GPU.
#include <algorithm>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "dv_internal.h"

struct dv_packed_region {
  std::vector<int32_t> item_variant_start, item_image_start, item_candidate;
  std::vector<uint32_t> item_ref_idx, item_list_off, item_combo, list_read;
  std::vector<uint16_t> item_height;
  std::vector<uint64_t> item_out_off;
  std::vector<uint8_t> list_code, list_group;
  uint32_t max_list_len = 0;
};

namespace {

inline uint64_t fnv1a(const char* s, size_t n, uint64_t h = 1469598103934665603ull) {
  for (size_t i = 0; i < n; ++i) {
    h ^= static_cast<unsigned char>(s[i]);
    h *= 1099511628211ull;
  }
  return h;
}

// hash of "<name>/<digit>" without building the string
inline uint64_t key_hash(const char* name, size_t n, unsigned read_number) {
  uint64_t h = fnv1a(name, n);
  const char tail[2] = {'/', static_cast<char>('0' + read_number)};
  return fnv1a(tail, 2, h);
}

}  // namespace

extern "C" {

int dv_pack_region(const dv_pack_reads* reads, const dv_pack_options* opt, int32_t n_candidates,
                   const dv_pack_candidate* cands, const uint32_t* combo_masks,
                   const char* support_keys, const uint32_t* support_key_off,
                   const uint8_t* support_alt, dv_packed_region** out) {
  if (!reads || !opt || !out || n_candidates < 0 || (n_candidates && !cands)) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_pack_region: bad argument");
  }
  if (reads->n_reads < 0 || opt->width < 3 || opt->pileup_height <= 0) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_pack_region: bad sizes");
  }
  const int32_t n = reads->n_reads;
  std::vector<uint64_t> rhash(n);
  std::vector<uint32_t> rlen(n);
  bool sorted = true;
  int64_t max_span = 1;
  {
    // key hashes of all reads: slices on the same number of host threads as the candidates
    const int nt = std::max(1, std::min(opt->n_threads > 0 ? opt->n_threads : 1, n / 4096));
    std::vector<int> bad(nt, 0);
    auto hash_slice = [&](int t) {
      const int32_t r0 = static_cast<int32_t>(static_cast<int64_t>(n) * t / nt);
      const int32_t r1 = static_cast<int32_t>(static_cast<int64_t>(n) * (t + 1) / nt);
      for (int32_t r = r0; r < r1; ++r) {
        const char* nm = reads->names + reads->name_off[r];
        rlen[r] = static_cast<uint32_t>(std::strlen(nm));
        if (reads->read_number[r] > 9) {
          bad[t] = 1;
          return;
        }
        rhash[r] = key_hash(nm, rlen[r], reads->read_number[r]);
      }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(hash_slice, t);
    hash_slice(0);
    for (std::thread& th : pool) th.join();
    for (int b : bad) {
      if (b) return dv::fail(DV_ERR_BAD_INPUT, "read_number > 9");
    }
  }
  for (int32_t r = 0; r < n; ++r) {
    if (r && reads->read_pos[r] < reads->read_pos[r - 1]) sorted = false;
    max_span = std::max<int64_t>(max_span, reads->read_end[r] - reads->read_pos[r]);
  }
  // Query by binary search over the start positions; reads that did not arrive sorted
  // (caller order is what the reference's Query returns) are searched through a sorted
  // index and the hits put back into caller order.
  std::vector<int32_t> order, pos_sorted;
  const int32_t* spos = reads->read_pos;
  if (!sorted) {
    order.resize(n);
    for (int32_t r = 0; r < n; ++r) order[r] = r;
    std::stable_sort(order.begin(), order.end(),
                     [&](int32_t a, int32_t b) { return reads->read_pos[a] < reads->read_pos[b]; });
    pos_sorted.resize(n);
    for (int32_t k = 0; k < n; ++k) pos_sorted[k] = reads->read_pos[order[k]];
    spos = pos_sorted.data();
  }
  // Candidates are independent: a contiguous slice per thread, each into its own
  // dv_packed_region, concatenated in candidate order (offsets shifted) afterwards.
  const int half = (opt->width - 1) / 2;
  struct Sup {
    uint32_t key_off, key_len;
    uint8_t first_alt, last_alt;
  };
  auto pack_slice = [&](int32_t c_begin, int32_t c_end, dv_packed_region* pr, std::string* error) {
    pr->item_list_off.push_back(0);
    std::unordered_multimap<uint64_t, Sup> sup;
    std::vector<uint32_t> picked;
    std::vector<uint8_t> first_alt, group;
    for (int32_t ci = c_begin; ci < c_end; ++ci) {
      const dv_pack_candidate& c = cands[ci];
      if (c.ref_idx < 0) continue;  // no reference window: the reference skips the candidate (:650-653)
      if (c.n_alts < 0 || c.n_alts > 32) {
        *error = "n_alts out of range";
        return;
      }
      // ---- allele_support of this candidate: key -> (first alt, last alt) ----
      sup.clear();
      for (uint32_t j = c.first_support; j < c.first_support + c.n_support; ++j) {
        const char* k = support_keys + support_key_off[j];
        const uint32_t kl = static_cast<uint32_t>(std::strlen(k));
        const uint64_t h = fnv1a(k, kl);
        const uint8_t alt = support_alt[j];
        if (alt >= c.n_alts) {
          *error = "support_alt out of range";
          return;
        }
        bool found = false;
        auto range = sup.equal_range(h);
        for (auto it = range.first; it != range.second; ++it) {
          if (it->second.key_len == kl && std::memcmp(support_keys + it->second.key_off, k, kl) == 0) {
            it->second.first_alt = std::min(it->second.first_alt, alt);
            it->second.last_alt = std::max(it->second.last_alt, alt);
            found = true;
            break;
          }
        }
        if (!found) sup.emplace(h, Sup{support_key_off[j], kl, alt, alt});
      }
      // ---- Query: reads overlapping the window, caller order ----
      const int64_t q0 = c.start - opt->read_overlap_buffer_bp;
      const int64_t q1 = c.end + opt->read_overlap_buffer_bp;
      picked.clear();
      const int32_t lo = static_cast<int32_t>(
          std::lower_bound(spos, spos + n,
                           static_cast<int32_t>(std::max<int64_t>(q0 - max_span + 1, INT32_MIN))) - spos);
      const int32_t hi = static_cast<int32_t>(
          std::lower_bound(spos + lo, spos + n, static_cast<int32_t>(std::min<int64_t>(q1, INT32_MAX))) - spos);
      for (int32_t k = lo; k < hi; ++k) {
        const int32_t r = sorted ? k : order[k];
        if (q1 > reads->read_pos[r] && q0 < reads->read_end[r]) picked.push_back(static_cast<uint32_t>(r));
      }
      if (!sorted) std::sort(picked.begin(), picked.end());
      first_alt.assign(picked.size(), 255);
      group.assign(picked.size(), static_cast<uint8_t>(c.n_alts));
      if (!sup.empty()) {
        for (size_t j = 0; j < picked.size(); ++j) {
          const uint32_t r = picked[j];
          auto range = sup.equal_range(rhash[r]);
          for (auto it = range.first; it != range.second; ++it) {
            const char* k = support_keys + it->second.key_off;
            const char* nm = reads->names + reads->name_off[r];
            if (it->second.key_len == rlen[r] + 2 && std::memcmp(k, nm, rlen[r]) == 0 &&
                k[rlen[r]] == '/' && k[rlen[r] + 1] == static_cast<char>('0' + reads->read_number[r])) {
              first_alt[j] = it->second.first_alt;
              group[j] = it->second.last_alt;
              break;
            }
          }
        }
      }
      // ---- one item per alt combination ----
      for (uint32_t k = 0; k < c.n_combos; ++k) {
        const uint32_t mask = combo_masks[c.first_combo + k];
        pr->item_variant_start.push_back(static_cast<int32_t>(c.start));
        pr->item_image_start.push_back(static_cast<int32_t>(c.start - half));
        pr->item_ref_idx.push_back(static_cast<uint32_t>(c.ref_idx));
        pr->item_height.push_back(static_cast<uint16_t>(opt->pileup_height));
        pr->item_candidate.push_back(ci);
        pr->item_combo.push_back(mask);
        for (size_t j = 0; j < picked.size(); ++j) {
          pr->list_read.push_back(picked[j]);
          pr->list_code.push_back(first_alt[j] == 255 ? 0 : ((mask >> first_alt[j]) & 1u) ? 1 : 2);
          pr->list_group.push_back(group[j]);
        }
        pr->item_list_off.push_back(static_cast<uint32_t>(pr->list_read.size()));
        pr->max_list_len = std::max<uint32_t>(pr->max_list_len, static_cast<uint32_t>(picked.size()));
      }
    }
  };
  int n_threads = opt->n_threads > 0 ? opt->n_threads : 1;
  n_threads = std::max(1, std::min(n_threads, n_candidates / 64));   // small regions: one thread
  std::vector<std::unique_ptr<dv_packed_region>> parts;
  std::vector<std::string> errors(n_threads);
  for (int t = 0; t < n_threads; ++t) parts.push_back(std::make_unique<dv_packed_region>());
  {
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) {
      pool.emplace_back(pack_slice, static_cast<int32_t>(static_cast<int64_t>(n_candidates) * t / n_threads),
                        static_cast<int32_t>(static_cast<int64_t>(n_candidates) * (t + 1) / n_threads),
                        parts[t].get(), &errors[t]);
    }
    pack_slice(0, static_cast<int32_t>(static_cast<int64_t>(n_candidates) / n_threads), parts[0].get(), &errors[0]);
    for (std::thread& th : pool) th.join();
  }
  for (const std::string& e : errors) {
    if (!e.empty()) return dv::fail(DV_ERR_INVALID_ARGUMENT, e);
  }
  std::unique_ptr<dv_packed_region> pr = std::move(parts[0]);
  for (int t = 1; t < n_threads; ++t) {
    const dv_packed_region& q = *parts[t];
    const uint32_t shift = static_cast<uint32_t>(pr->list_read.size());
    auto append = [](auto& dst, const auto& src) { dst.insert(dst.end(), src.begin(), src.end()); };
    append(pr->item_variant_start, q.item_variant_start);
    append(pr->item_image_start, q.item_image_start);
    append(pr->item_ref_idx, q.item_ref_idx);
    append(pr->item_height, q.item_height);
    append(pr->item_candidate, q.item_candidate);
    append(pr->item_combo, q.item_combo);
    append(pr->list_read, q.list_read);
    append(pr->list_code, q.list_code);
    append(pr->list_group, q.list_group);
    for (size_t k = 1; k < q.item_list_off.size(); ++k) pr->item_list_off.push_back(q.item_list_off[k] + shift);
    pr->max_list_len = std::max(pr->max_list_len, q.max_list_len);
  }
  pr->item_out_off.resize(pr->item_height.size());
  for (size_t item = 0; item < pr->item_out_off.size(); ++item) {
    pr->item_out_off[item] = static_cast<uint64_t>(item) * opt->example_bytes;
  }
  *out = pr.release();
  return DV_OK;
}

int dv_packed_region_fill_batch(const dv_packed_region* p, int use_groups, dv_batch* b) {
  if (!p || !b) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_packed_region_fill_batch: null");
  b->n_items = static_cast<int32_t>(p->item_height.size());
  b->n_list = static_cast<uint32_t>(p->list_read.size());
  b->max_list_len = p->max_list_len;
  b->item_variant_start = p->item_variant_start.data();
  b->item_image_start = p->item_image_start.data();
  b->item_ref_idx = p->item_ref_idx.data();
  b->item_list_off = p->item_list_off.data();
  b->item_height = p->item_height.data();
  b->item_out_off = p->item_out_off.data();
  b->list_read = p->list_read.data();
  b->list_code = p->list_code.data();
  b->list_group = use_groups ? p->list_group.data() : nullptr;
  return DV_OK;
}

int dv_packed_region_items(const dv_packed_region* p, const int32_t** item_candidate,
                           const uint32_t** item_combo) {
  if (!p) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_packed_region_items: null");
  if (item_candidate) *item_candidate = p->item_candidate.data();
  if (item_combo) *item_combo = p->item_combo.data();
  return static_cast<int>(p->item_height.size());
}

void dv_packed_region_free(dv_packed_region* p) { delete p; }

}  // extern "C"
