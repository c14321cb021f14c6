// TEST INFRASTRUCTURE (oracle/ref_build): absl::flat_hash_map = std::unordered_map.
#ifndef DVREF_ABSL_FLAT_HASH_MAP_H_
#define DVREF_ABSL_FLAT_HASH_MAP_H_
#include <unordered_map>
namespace absl {
template <class K, class V, class H = std::hash<K>, class E = std::equal_to<K>>
using flat_hash_map = std::unordered_map<K, V, H, E>;
template <class K, class V, class H = std::hash<K>, class E = std::equal_to<K>>
using node_hash_map = std::unordered_map<K, V, H, E>;
}
#endif
