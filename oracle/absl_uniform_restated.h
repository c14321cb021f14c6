// absl_uniform_restated.h -- TEST INFRASTRUCTURE.  absl::Uniform<IntType>(absl::IntervalClosed, gen, lo, hi) over a
// STANDARD 64-bit generator (std::mt19937_64), restated from abseil's published sources because abseil itself is not in
// this image (the reference pins it through its bazel WORKSPACE; nothing under /root/reference vendors it):
//   absl/random/distributions.h      Uniform(tag, urbg, lo, hi) -> uniform_int_distribution<T>(lo', hi')(urbg) for a
//                                    non-abseil URBG, the closed interval taken as it stands
//   absl/random/uniform_int_distribution.h   operator(): a + Generate(g, b - a);  Generate(g, R):
//         bits = FastUniformBits<uint64>(g)        -- for a generator whose range is the full 64 bits: g() itself
//         Lim = R + 1;  if ((R & Lim) == 0) return bits & R          -- power-of-two ranges: the low bits
//         product = bits * Lim (128 bit);  if (lo64(product) < Lim) { threshold = (2^64 - Lim) % Lim;
//             while (lo64(product) < threshold) { bits = g(); product = bits * Lim; } }
//         return hi64(product)                                        -- Lemire's multiply-and-reject
// PARITY UNPINNED for this one function: no reference test fixes a draw of absl::Uniform (sampling_util_test.cc injects
// its own index providers), so what is pinned is everything around it.  Used by the oracle restatement and by the
// reference build's absl stand-in, so that the two can be compared through the non-uniform downsampling path.
#ifndef DVO_ABSL_UNIFORM_RESTATED_H_
#define DVO_ABSL_UNIFORM_RESTATED_H_
#include <cstdint>

namespace dvo_absl {

template <class G>
inline uint64_t UniformClosed64(G& g, uint64_t lo, uint64_t hi) {
  const uint64_t R = hi - lo;
  uint64_t bits = static_cast<uint64_t>(g());
  const uint64_t Lim = R + 1;
  if ((R & Lim) == 0) return lo + (bits & R);
  unsigned __int128 product = static_cast<unsigned __int128>(bits) * Lim;
  if (static_cast<uint64_t>(product) < Lim) {
    const uint64_t threshold = (~static_cast<uint64_t>(0) - Lim + 1) % Lim;
    while (static_cast<uint64_t>(product) < threshold) {
      bits = static_cast<uint64_t>(g());
      product = static_cast<unsigned __int128>(bits) * Lim;
    }
  }
  return lo + static_cast<uint64_t>(product >> 64);
}

}  // namespace dvo_absl
#endif  // DVO_ABSL_UNIFORM_RESTATED_H_
