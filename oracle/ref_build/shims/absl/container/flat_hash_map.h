// TEST INFRASTRUCTURE (oracle/ref_build): absl::flat_hash_map = std::unordered_map.
#ifndef DVREF_ABSL_FLAT_HASH_MAP_H_
#define DVREF_ABSL_FLAT_HASH_MAP_H_
#include <unordered_map>
#include "absl/hash/hash.h"
namespace absl {
template <class K, class V, class H = absl::Hash<K>, class E = absl::dvref_hash::DefaultEq<K>>
using flat_hash_map = std::unordered_map<K, V, H, E>;
template <class K, class V, class H = absl::Hash<K>, class E = absl::dvref_hash::DefaultEq<K>>
using node_hash_map = std::unordered_map<K, V, H, E>;
}
#endif
