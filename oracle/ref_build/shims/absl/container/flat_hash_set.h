// TEST INFRASTRUCTURE (oracle/ref_build): absl::flat_hash_set = std::unordered_set (C++20: contains()).
#ifndef DVREF_ABSL_FLAT_HASH_SET_H_
#define DVREF_ABSL_FLAT_HASH_SET_H_
#include <unordered_set>
#include "absl/hash/hash.h"
namespace absl {
template <class T, class H = absl::Hash<T>, class E = absl::dvref_hash::DefaultEq<T>>
using flat_hash_set = std::unordered_set<T, H, E>;
template <class T, class H = absl::Hash<T>, class E = absl::dvref_hash::DefaultEq<T>>
using node_hash_set = std::unordered_set<T, H, E>;
}
#endif
