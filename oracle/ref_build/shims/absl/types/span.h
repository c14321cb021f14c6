// TEST INFRASTRUCTURE (oracle/ref_build): absl::Span on plain pointers (implicit from containers, as abseil's).
#ifndef DVREF_ABSL_SPAN_H_
#define DVREF_ABSL_SPAN_H_
#include <cstddef>
#include <initializer_list>
#include <type_traits>
#include <vector>
namespace absl {
template <class T>
class Span {
 public:
  using value_type = std::remove_cv_t<T>;
  using iterator = T*;
  using const_iterator = const T*;
  constexpr Span() : p_(nullptr), n_(0) {}
  constexpr Span(T* p, size_t n) : p_(p), n_(n) {}
  template <size_t N>
  constexpr Span(T (&a)[N]) : p_(a), n_(N) {}
  template <class C, class = decltype(std::declval<C&>().data()), class = decltype(std::declval<C&>().size()),
            class = std::enable_if_t<std::is_convertible_v<decltype(std::declval<C&>().data()), T*>>>
  constexpr Span(C& c) : p_(c.data()), n_(c.size()) {}
  template <class C, class = decltype(std::declval<const C&>().data()), class = decltype(std::declval<const C&>().size()),
            class = std::enable_if_t<std::is_const_v<T> && std::is_convertible_v<decltype(std::declval<const C&>().data()), T*>>>
  constexpr Span(const C& c) : p_(c.data()), n_(c.size()) {}
  template <class U = T, class = std::enable_if_t<std::is_const_v<U>>>
  Span(std::initializer_list<value_type> l) : p_(l.begin()), n_(l.size()) {}
  constexpr T* data() const { return p_; }
  constexpr size_t size() const { return n_; }
  constexpr size_t length() const { return n_; }
  constexpr bool empty() const { return n_ == 0; }
  constexpr T& operator[](size_t i) const { return p_[i]; }
  constexpr T& at(size_t i) const { return p_[i]; }
  constexpr T& front() const { return p_[0]; }
  constexpr T& back() const { return p_[n_ - 1]; }
  constexpr T* begin() const { return p_; }
  constexpr T* end() const { return p_ + n_; }
  constexpr const T* cbegin() const { return p_; }
  constexpr const T* cend() const { return p_ + n_; }
  constexpr Span subspan(size_t pos, size_t len = static_cast<size_t>(-1)) const {
    return Span(p_ + pos, len < n_ - pos ? len : n_ - pos);
  }
 private:
  T* p_;
  size_t n_;
};
template <class C>
Span<const typename C::value_type> MakeConstSpan(const C& c) { return Span<const typename C::value_type>(c.data(), c.size()); }
template <class C>
Span<typename C::value_type> MakeSpan(C& c) { return Span<typename C::value_type>(c.data(), c.size()); }
}  // namespace absl
#endif
