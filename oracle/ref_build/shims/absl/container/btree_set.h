// TEST INFRASTRUCTURE (oracle/ref_build): absl::btree_set = std::set (same ordered iteration).
#ifndef DVREF_ABSL_BTREE_SET_H_
#define DVREF_ABSL_BTREE_SET_H_
#include <set>
namespace absl {
template <class T, class C = std::less<T>>
using btree_set = std::set<T, C>;
}
#endif
