// TEST INFRASTRUCTURE (oracle/ref_build): absl::Uniform over a standard generator.  NOT abseil's bit stream: the one
// caller in the encoder (non-uniform downsampling, DeepSomatic's tumour sample) therefore draws other reads here than a
// real build would; nothing compares that path (the product refuses it).
#ifndef DVREF_ABSL_RANDOM_H_
#define DVREF_ABSL_RANDOM_H_
#include <random>
namespace absl {
struct IntervalClosedTag {};
struct IntervalClosedOpenTag {};
inline constexpr IntervalClosedTag IntervalClosed{};
inline constexpr IntervalClosedOpenTag IntervalClosedOpen{};
using BitGen = std::mt19937_64;
template <class T, class G, class A, class B>
T Uniform(IntervalClosedTag, G& g, A lo, B hi) { return std::uniform_int_distribution<T>(static_cast<T>(lo), static_cast<T>(hi))(g); }
template <class T, class G, class A, class B>
T Uniform(IntervalClosedOpenTag, G& g, A lo, B hi) { return std::uniform_int_distribution<T>(static_cast<T>(lo), static_cast<T>(hi) - 1)(g); }
template <class T, class G, class A, class B>
T Uniform(G& g, A lo, B hi) { return std::uniform_int_distribution<T>(static_cast<T>(lo), static_cast<T>(hi) - 1)(g); }
}  // namespace absl
#endif
