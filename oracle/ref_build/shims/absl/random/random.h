// TEST INFRASTRUCTURE (oracle/ref_build): absl::Uniform over a standard generator.  The closed-interval form -- the one
// the encoder's non-uniform downsampling calls (deepvariant/sampling_util.h:129-137) -- follows abseil's published
// algorithm as restated in oracle/absl_uniform_restated.h (abseil is not in this image: PARITY UNPINNED for the bit
// stream itself, see there); the other forms, which nothing on the path calls, draw from the standard library.
#ifndef DVREF_ABSL_RANDOM_H_
#define DVREF_ABSL_RANDOM_H_
#include <random>

#include "absl_uniform_restated.h"
namespace absl {
struct IntervalClosedTag {};
struct IntervalClosedOpenTag {};
inline constexpr IntervalClosedTag IntervalClosed{};
inline constexpr IntervalClosedOpenTag IntervalClosedOpen{};
using BitGen = std::mt19937_64;
template <class T, class G, class A, class B>
T Uniform(IntervalClosedTag, G& g, A lo, B hi) {
  return static_cast<T>(dvo_absl::UniformClosed64(g, static_cast<uint64_t>(static_cast<T>(lo)), static_cast<uint64_t>(static_cast<T>(hi))));
}
template <class T, class G, class A, class B>
T Uniform(IntervalClosedOpenTag, G& g, A lo, B hi) { return std::uniform_int_distribution<T>(static_cast<T>(lo), static_cast<T>(hi) - 1)(g); }
template <class T, class G, class A, class B>
T Uniform(G& g, A lo, B hi) { return std::uniform_int_distribution<T>(static_cast<T>(lo), static_cast<T>(hi) - 1)(g); }
}  // namespace absl
#endif
