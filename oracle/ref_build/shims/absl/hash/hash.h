// TEST INFRASTRUCTURE (oracle/ref_build): absl::Hash<T> for what the compiled sources key their tables with --
// anything std::hash takes, enums, pairs, tuples, and combinations of those.
#ifndef DVREF_ABSL_HASH_H_
#define DVREF_ABSL_HASH_H_
#include <cstddef>
#include <functional>
#include <string>
#include <string_view>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>
namespace absl {
namespace dvref_hash {
inline size_t Mix(size_t seed, size_t v) { return seed ^ (v + 0x9e3779b97f4a7c15ull + (seed << 6) + (seed >> 2)); }
template <class T, class = void>
struct H {
  size_t operator()(const T& v) const { return std::hash<T>()(v); }
};
// strings hash transparently: find(string_view) / find(const char*) on a table keyed by std::string
template <>
struct H<std::string, void> {
  using is_transparent = void;
  size_t operator()(std::string_view v) const { return std::hash<std::string_view>()(v); }
};
template <>
struct H<std::string_view, void> {
  using is_transparent = void;
  size_t operator()(std::string_view v) const { return std::hash<std::string_view>()(v); }
};
template <class K>
using DefaultEq = std::conditional_t<std::is_same_v<K, std::string> || std::is_same_v<K, std::string_view>, std::equal_to<>,
                                     std::equal_to<K>>;
template <class T>
struct H<T, std::enable_if_t<std::is_enum_v<T>>> {
  size_t operator()(const T& v) const { return std::hash<long long>()(static_cast<long long>(v)); }
};
template <class A, class B>
struct H<std::pair<A, B>, void> {
  size_t operator()(const std::pair<A, B>& p) const { return Mix(H<A>()(p.first), H<B>()(p.second)); }
};
template <class... T>
struct H<std::tuple<T...>, void> {
  size_t operator()(const std::tuple<T...>& t) const {
    size_t seed = 0;
    std::apply([&](const auto&... v) { ((seed = Mix(seed, H<std::decay_t<decltype(v)>>()(v))), ...); }, t);
    return seed;
  }
};
template <class T>
struct H<std::vector<T>, void> {
  size_t operator()(const std::vector<T>& v) const {
    size_t seed = v.size();
    for (const auto& e : v) seed = Mix(seed, H<T>()(e));
    return seed;
  }
};
// types that hash themselves the abseil way: friend H AbslHashValue(H h, const T&) { return H::combine(std::move(h), ...); }
struct HashState {
  size_t v = 0;
  template <class... A>
  static HashState combine(HashState h, const A&... a) {
    ((h.v = Mix(h.v, H<A>()(a))), ...);
    return h;
  }
};
template <class T>
struct H<T, std::void_t<decltype(AbslHashValue(std::declval<HashState>(), std::declval<const T&>()))>> {
  size_t operator()(const T& v) const { return AbslHashValue(HashState{}, v).v; }
};
}  // namespace dvref_hash
template <class T>
using Hash = dvref_hash::H<T>;
}  // namespace absl
#endif
