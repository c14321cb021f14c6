// TEST INFRASTRUCTURE (oracle/ref_build): the absl::c_* algorithms the reference's encoder uses.
#ifndef DVREF_ABSL_ALGORITHM_CONTAINER_H_
#define DVREF_ABSL_ALGORITHM_CONTAINER_H_
#include <algorithm>
#include <numeric>
namespace absl {
template <class C, class Cmp> void c_stable_sort(C& c, Cmp cmp) { std::stable_sort(c.begin(), c.end(), cmp); }
template <class C> void c_stable_sort(C& c) { std::stable_sort(c.begin(), c.end()); }
template <class C, class Cmp> void c_sort(C& c, Cmp cmp) { std::sort(c.begin(), c.end(), cmp); }
template <class C> void c_sort(C& c) { std::sort(c.begin(), c.end()); }
template <class C, class T> auto c_find(C& c, const T& v) { return std::find(c.begin(), c.end(), v); }
template <class C, class T> bool c_linear_search(const C& c, const T& v) { return std::find(c.begin(), c.end(), v) != c.end(); }
template <class C, class T> void c_iota(C& c, T v) { std::iota(c.begin(), c.end(), v); }
template <class C, class P> bool c_any_of(const C& c, P p) { return std::any_of(c.begin(), c.end(), p); }
template <class C, class P> bool c_all_of(const C& c, P p) { return std::all_of(c.begin(), c.end(), p); }
template <class C, class G> void c_shuffle(C& c, G&& g) { std::shuffle(c.begin(), c.end(), g); }
}
#endif
