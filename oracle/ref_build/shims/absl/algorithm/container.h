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
template <class C, class P> auto c_count_if(const C& c, P p) { return std::count_if(c.begin(), c.end(), p); }
template <class C, class T> auto c_count(const C& c, const T& v) { return std::count(c.begin(), c.end(), v); }
template <class C, class P> auto c_find_if(C& c, P p) { return std::find_if(c.begin(), c.end(), p); }
template <class C, class P> bool c_none_of(const C& c, P p) { return std::none_of(c.begin(), c.end(), p); }
template <class C> auto c_max_element(C& c) { return std::max_element(c.begin(), c.end()); }
template <class C, class Cmp> auto c_max_element(C& c, Cmp cmp) { return std::max_element(c.begin(), c.end(), cmp); }
template <class C> auto c_min_element(C& c) { return std::min_element(c.begin(), c.end()); }
template <class C, class T> T c_accumulate(const C& c, T init) { return std::accumulate(c.begin(), c.end(), init); }
template <class C, class O> O c_copy(const C& c, O out) { return std::copy(c.begin(), c.end(), out); }
template <class C> void c_reverse(C& c) { std::reverse(c.begin(), c.end()); }
}
#endif
