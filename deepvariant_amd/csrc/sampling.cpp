// sampling.cpp -- SampleOptions.use_non_uniform_downsampling on the host: which reads of a pile-up survive when every
// allele keeps a minimum of its supporters (DeepSomatic's tumour sample; a general make_examples flag).
//   deepvariant/pileup_image_native.cc:242-294  GetReadIndicesAllelePartition, DownsampleReadIndicesWithMinsPerAllele
//   deepvariant/sampling_util.h:57-155          ReservoirSampleImpl, SampleWithPartitionMinsImpl, ReservoirSample
// The draws are absl::Uniform<size_t>(absl::IntervalClosed, gen, 0, index) over the std::mt19937_64 seeded with
// PileupImageOptions.random_seed.  abseil is a third-party dependency the reference does not vendor (bazel WORKSPACE);
// its algorithm for a standard 64-bit generator is restated below from abseil's published sources
// (absl/random/uniform_int_distribution.h: power-of-two ranges take the low bits, every other range Lemire's
// multiply-and-reject on the 128-bit product).  No reference test fixes a draw (sampling_util_test.cc injects its own
// index providers), so for that one function parity is UNPINNED; everything around it is pinned -- by the reference's
// two distribution tests, re-run here through `forced_draws`, and against the reference's own code compiled in
// oracle/_ref (which draws through the same restatement).
// The sample is a SET in the reference: the encoder then walks it in ascending read order, so the host hands the
// device a shorter read list and the device's own shuffle has nothing left to do.
#include <algorithm>
#include <cstdint>
#include <random>
#include <set>
#include <vector>

#include "dv_internal.h"
#include "dvhip.h"

namespace {

struct Draws {
  std::mt19937_64 gen;
  const uint64_t* forced;
  int64_t n_forced, at = 0;
  bool exhausted = false;

  // absl::Uniform<size_t>(absl::IntervalClosed, gen, 0, max)
  uint64_t closed(uint64_t max) {
    if (forced) {
      if (at >= n_forced) {
        exhausted = true;
        return 0;
      }
      return forced[at++];
    }
    const uint64_t R = max;
    uint64_t bits = gen();
    const uint64_t Lim = R + 1;
    if ((R & Lim) == 0) return bits & R;
    unsigned __int128 product = static_cast<unsigned __int128>(bits) * Lim;
    if (static_cast<uint64_t>(product) < Lim) {
      const uint64_t threshold = (~static_cast<uint64_t>(0) - Lim + 1) % Lim;
      while (static_cast<uint64_t>(product) < threshold) {
        bits = gen();
        product = static_cast<unsigned __int128>(bits) * Lim;
      }
    }
    return static_cast<uint64_t>(product >> 64);
  }
};

// ReservoirSample(sample_size, gen, population): a draw is made for every element beyond the first sample_size, also
// when sample_size is 0.
std::set<int32_t> reservoir(Draws* d, const std::set<int32_t>& population, int sample_size) {
  if (population.size() < static_cast<size_t>(sample_size)) return population;
  std::vector<int32_t> sampled(static_cast<size_t>(sample_size));
  size_t index = 0;
  auto it = population.begin();
  for (; index < static_cast<size_t>(sample_size); ++it, ++index) sampled[index] = *it;
  for (; it != population.end(); ++it, ++index) {
    const uint64_t swap_index = d->closed(index);
    if (swap_index < static_cast<uint64_t>(sample_size)) sampled[swap_index] = *it;
  }
  return std::set<int32_t>(sampled.begin(), sampled.end());
}

}  // namespace

extern "C" int dv_downsample_with_partition_mins(int32_t n_reads, const int32_t* part_off, const int32_t* part_idx,
                                                 int32_t n_parts, int32_t max_reads, int32_t min_per_partition,
                                                 uint32_t random_seed, const uint64_t* forced_draws, int64_t n_forced,
                                                 int32_t* out, int32_t* n_out) {
  if (n_reads < 0 || n_parts < 0 || max_reads < 0 || !n_out || (n_reads > 0 && !out) || (n_parts > 0 && (!part_off || (part_off[n_parts] > 0 && !part_idx)))) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_downsample_with_partition_mins: bad argument");
  }
  // GetReadIndicesAllelePartition: a read belongs to the first element that lists it.  The caller lists the alleles'
  // supporters and, as the last element, the reads no allele lists; a read in NO element cannot be drawn (the reference
  // finds reads through a name -> index map, so of several reads with one key only the last is ever seen).  The
  // partition is a SET of sets: equal elements collapse (several alleles without reads), and it is walked in the
  // sets' lexicographic order.
  std::vector<char> taken(static_cast<size_t>(n_reads), 0);
  std::set<std::set<int32_t>> partition;
  for (int32_t p = 0; p < n_parts; ++p) {
    std::set<int32_t> element;
    for (int32_t k = part_off[p]; k < part_off[p + 1]; ++k) {
      const int32_t r = part_idx[k];
      if (r < 0 || r >= n_reads) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_downsample_with_partition_mins: read index out of range");
      if (!taken[static_cast<size_t>(r)]) {
        taken[static_cast<size_t>(r)] = 1;
        element.insert(r);
      }
    }
    partition.insert(std::move(element));
  }

  Draws draws{std::mt19937_64(random_seed), forced_draws, n_forced};
  std::set<int32_t> sampled, unsampled;
  for (const std::set<int32_t>& element : partition) {   // SampleWithPartitionMinsImpl
    const std::set<int32_t> chosen = reservoir(&draws, element, min_per_partition);
    for (int32_t e : element) {
      if (!chosen.count(e)) unsampled.insert(e);
    }
    sampled.insert(chosen.begin(), chosen.end());
  }
  const int64_t remaining = static_cast<int64_t>(max_reads) - static_cast<int64_t>(sampled.size());
  if (remaining < 0) {   // "Threshold of N per partition results in more than sample_size elements": the caller
    *n_out = -1;         // falls back to the uniform shuffle (pileup_image_native.cc:333-337)
    return DV_OK;
  }
  const std::set<int32_t> more = reservoir(&draws, unsampled, static_cast<int>(remaining));
  sampled.insert(more.begin(), more.end());
  if (draws.exhausted) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_downsample_with_partition_mins: forced_draws ran out");
  int32_t n = 0;
  for (int32_t e : sampled) out[n++] = e;
  *n_out = n;
  return DV_OK;
}
