// TEST INFRASTRUCTURE (oracle/ref_build): absl::btree_map = std::map.
#ifndef DVREF_ABSL_BTREE_MAP_H_
#define DVREF_ABSL_BTREE_MAP_H_
#include <map>
namespace absl {
template <class K, class V, class C = std::less<K>>
using btree_map = std::map<K, V, C>;
}
#endif
