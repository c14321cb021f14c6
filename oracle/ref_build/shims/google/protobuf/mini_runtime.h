// mini_runtime.h -- the container types behind the structs oracle/ref_build/mini_protoc.py generates.
// TEST INFRASTRUCTURE (oracle/): stands in for libprotobuf's RepeatedField / RepeatedPtrField / Map so that the
// reference's sources compile unmodified.  Only the interface those sources use; same semantics
// (int sizes, `at` on a missing key is fatal, default instances for unset message fields).
#ifndef GOOGLE_PROTOBUF_MINI_RUNTIME_H_
#define GOOGLE_PROTOBUF_MINI_RUNTIME_H_

#include <stdexcept>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <initializer_list>
#include <iterator>
#include <map>
#include <type_traits>
#include <memory>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

namespace google {
namespace protobuf {

template <class T>
class RepeatedField : public std::vector<T> {
 public:
  using std::vector<T>::vector;
  int size() const { return static_cast<int>(std::vector<T>::size()); }
  T Get(int i) const { return (*this)[i]; }
  void Set(int i, T v) { (*this)[i] = v; }
  void Add(T v) { this->push_back(v); }
  void Clear() { this->clear(); }
  void Reserve(int n) { this->reserve(n); }
  T* mutable_data() { return this->data(); }
};

// Elements are held through pointers, as in libprotobuf: pointer_begin() / pointer_end() expose the pointer array
// (the reference sorts alternate_bases through it), everything else sees T&.
template <class T>
class RepeatedPtrField {
  template <bool Const>
  class Iter {
   public:
    using iterator_category = std::random_access_iterator_tag;
    using value_type = T;
    using difference_type = std::ptrdiff_t;
    using pointer = std::conditional_t<Const, const T*, T*>;
    using reference = std::conditional_t<Const, const T&, T&>;
    Iter() : p_(nullptr) {}
    explicit Iter(T* const* p) : p_(p) {}
    template <bool C = Const, class = std::enable_if_t<C>>
    Iter(const Iter<false>& o) : p_(o.raw()) {}
    reference operator*() const { return **p_; }
    pointer operator->() const { return *p_; }
    reference operator[](difference_type n) const { return *p_[n]; }
    Iter& operator++() { ++p_; return *this; }
    Iter operator++(int) { Iter t = *this; ++p_; return t; }
    Iter& operator--() { --p_; return *this; }
    Iter operator--(int) { Iter t = *this; --p_; return t; }
    Iter& operator+=(difference_type n) { p_ += n; return *this; }
    Iter& operator-=(difference_type n) { p_ -= n; return *this; }
    friend Iter operator+(Iter a, difference_type n) { return a += n; }
    friend Iter operator+(difference_type n, Iter a) { return a += n; }
    friend Iter operator-(Iter a, difference_type n) { return a -= n; }
    friend difference_type operator-(const Iter& a, const Iter& b) { return a.p_ - b.p_; }
    friend bool operator==(const Iter& a, const Iter& b) { return a.p_ == b.p_; }
    friend bool operator!=(const Iter& a, const Iter& b) { return a.p_ != b.p_; }
    friend bool operator<(const Iter& a, const Iter& b) { return a.p_ < b.p_; }
    friend bool operator>(const Iter& a, const Iter& b) { return a.p_ > b.p_; }
    friend bool operator<=(const Iter& a, const Iter& b) { return a.p_ <= b.p_; }
    friend bool operator>=(const Iter& a, const Iter& b) { return a.p_ >= b.p_; }
    T* const* raw() const { return p_; }
   private:
    T* const* p_;
  };

 public:
  using value_type = T;
  using iterator = Iter<false>;
  using const_iterator = Iter<true>;
  using pointer_iterator = T**;
  using const_pointer_iterator = const T* const*;
  using size_type = int;
  using reference = T&;
  using const_reference = const T&;
  RepeatedPtrField() = default;
  RepeatedPtrField(const RepeatedPtrField& o) { for (const T* e : o.v_) v_.push_back(new T(*e)); }
  RepeatedPtrField(RepeatedPtrField&& o) noexcept : v_(std::move(o.v_)) { o.v_.clear(); }
  template <class It>
  RepeatedPtrField(It b, It e) { for (; b != e; ++b) v_.push_back(new T(*b)); }
  RepeatedPtrField(std::initializer_list<T> l) { for (const T& e : l) v_.push_back(new T(e)); }
  ~RepeatedPtrField() { Clear(); }
  RepeatedPtrField& operator=(const RepeatedPtrField& o) {
    if (this != &o) {
      Clear();
      for (const T* e : o.v_) v_.push_back(new T(*e));
    }
    return *this;
  }
  RepeatedPtrField& operator=(RepeatedPtrField&& o) noexcept {
    if (this != &o) {
      Clear();
      v_ = std::move(o.v_);
      o.v_.clear();
    }
    return *this;
  }
  int size() const { return static_cast<int>(v_.size()); }
  bool empty() const { return v_.empty(); }
  const T& Get(int i) const { return *v_[static_cast<size_t>(i)]; }
  T* Mutable(int i) { return v_[static_cast<size_t>(i)]; }
  const T& operator[](int i) const { return *v_[static_cast<size_t>(i)]; }
  T& operator[](int i) { return *v_[static_cast<size_t>(i)]; }
  const T& at(int i) const { return *v_.at(static_cast<size_t>(i)); }
  T& at(int i) { return *v_.at(static_cast<size_t>(i)); }
  T* Add() {
    v_.push_back(new T());
    return v_.back();
  }
  void Add(T&& v) { v_.push_back(new T(std::move(v))); }
  void Add(const T& v) { v_.push_back(new T(v)); }
  template <class It>
  void Add(It b, It e) { for (; b != e; ++b) v_.push_back(new T(*b)); }
  void push_back(const T& v) { v_.push_back(new T(v)); }
  void push_back(T&& v) { v_.push_back(new T(std::move(v))); }
  template <class... A>
  T& emplace_back(A&&... a) {
    v_.push_back(new T(std::forward<A>(a)...));
    return *v_.back();
  }
  T& back() { return *v_.back(); }
  const T& back() const { return *v_.back(); }
  T& front() { return *v_.front(); }
  const T& front() const { return *v_.front(); }
  void RemoveLast() {
    delete v_.back();
    v_.pop_back();
  }
  void DeleteSubrange(int start, int num) {
    for (int i = start; i < start + num; ++i) delete v_[static_cast<size_t>(i)];
    v_.erase(v_.begin() + start, v_.begin() + start + num);
  }
  iterator erase(const_iterator pos) {
    const std::ptrdiff_t i = pos.raw() - v_.data();
    DeleteSubrange(static_cast<int>(i), 1);
    return iterator(v_.data() + i);
  }
  void SwapElements(int a, int b) { std::swap(v_[static_cast<size_t>(a)], v_[static_cast<size_t>(b)]); }
  void Clear() {
    for (T* e : v_) delete e;
    v_.clear();
  }
  void clear() { Clear(); }
  void Reserve(int n) { v_.reserve(static_cast<size_t>(n)); }
  void reserve(size_t n) { v_.reserve(n); }
  void Swap(RepeatedPtrField* o) { v_.swap(o->v_); }
  void CopyFrom(const RepeatedPtrField& o) { *this = o; }
  void MergeFrom(const RepeatedPtrField& o) { for (const T* e : o.v_) v_.push_back(new T(*e)); }
  iterator begin() { return iterator(v_.data()); }
  iterator end() { return iterator(v_.data() + v_.size()); }
  const_iterator begin() const { return const_iterator(v_.data()); }
  const_iterator end() const { return const_iterator(v_.data() + v_.size()); }
  const_iterator cbegin() const { return begin(); }
  const_iterator cend() const { return end(); }
  pointer_iterator pointer_begin() { return v_.data(); }
  pointer_iterator pointer_end() { return v_.data() + v_.size(); }
  const_pointer_iterator pointer_begin() const { return v_.data(); }
  const_pointer_iterator pointer_end() const { return v_.data() + v_.size(); }
  friend bool operator==(const RepeatedPtrField& a, const RepeatedPtrField& b) {
    if (a.size() != b.size()) return false;
    for (int i = 0; i < a.size(); ++i) {
      if (!(a[i] == b[i])) return false;
    }
    return true;
  }

 private:
  std::vector<T*> v_;
};

// google::protobuf::Map iterates in an unspecified (hash) order; code whose output depends on that order has no
// defined output in the reference either.  Here: key order.
template <class K, class V>
class Map : public std::map<K, V> {
 public:
  using std::map<K, V>::map;
  using Base = std::map<K, V>;
  bool contains(const K& k) const { return Base::find(k) != Base::end(); }
  template <class Q>
  bool contains(const Q& k) const { return Base::find(K(k)) != Base::end(); }
  template <class Q>
  const V& at(const Q& k) const {
    auto it = Base::find(K(k));
    if (it == Base::end()) {
      throw std::out_of_range("Check failed: key not found in protobuf Map::at");
    }
    return it->second;
  }
  template <class Q>
  V& at(const Q& k) {
    auto it = Base::find(K(k));
    if (it == Base::end()) {
      throw std::out_of_range("Check failed: key not found in protobuf Map::at");
    }
    return it->second;
  }
  template <class Q>
  typename Base::const_iterator find(const Q& k) const { return Base::find(K(k)); }
  template <class Q>
  typename Base::iterator find(const Q& k) { return Base::find(K(k)); }
  template <class Q>
  size_t count(const Q& k) const { return Base::count(K(k)); }
  template <class Q>
  V& operator[](const Q& k) { return Base::operator[](K(k)); }
};

namespace mini {

// wire-format primitives for the generated AppendTo()
inline void PutVarint(std::string* out, uint64_t v) {
  while (v >= 0x80) {
    out->push_back(static_cast<char>((v & 0x7F) | 0x80));
    v >>= 7;
  }
  out->push_back(static_cast<char>(v));
}
inline uint64_t ZigZag(int64_t v) { return (static_cast<uint64_t>(v) << 1) ^ static_cast<uint64_t>(v >> 63); }
template <class T>
inline void PutFixed(std::string* out, T v) {
  char b[sizeof(T)];
  std::memcpy(b, &v, sizeof(T));      // little-endian hosts only (x86-64)
  out->append(b, sizeof(T));
}
inline void PutBytes(std::string* out, const std::string& v) {
  PutVarint(out, v.size());
  out->append(v);
}

// what the reference asks an enum's descriptor: value_count(), value(i)->number() / ->name()
class EnumValueDescriptor {
 public:
  EnumValueDescriptor(const char* name, int number) : name_(name), number_(number) {}
  int number() const { return number_; }
  const std::string& name() const { return name_; }
 private:
  std::string name_;
  int number_;
};
class EnumDescriptor {
 public:
  EnumDescriptor(std::initializer_list<EnumValueDescriptor> v) : values_(v) {}
  int value_count() const { return static_cast<int>(values_.size()); }
  const EnumValueDescriptor* value(int i) const { return &values_[static_cast<size_t>(i)]; }
 private:
  std::vector<EnumValueDescriptor> values_;
};

inline const std::string& EmptyString() {
  static const std::string s;
  return s;
}

template <class T>
const T& Default() {
  static const T* d = new T();
  return *d;
}

// A singular message field: absent until mutable_get(), copied deeply, reads as the default instance.
template <class T>
class Box {
 public:
  Box() = default;
  Box(const Box& o) : p_(o.p_ ? new T(*o.p_) : nullptr) {}
  Box(Box&& o) noexcept = default;
  Box& operator=(const Box& o) {
    if (this != &o) p_.reset(o.p_ ? new T(*o.p_) : nullptr);
    return *this;
  }
  Box& operator=(Box&& o) noexcept = default;
  bool has() const { return p_ != nullptr; }
  const T& get() const { return p_ ? *p_ : Default<T>(); }
  T* mutable_get() {
    if (!p_) p_.reset(new T());
    return p_.get();
  }
  void reset() { p_.reset(); }
  void adopt(T* p) { p_.reset(p); }      // set_allocated_*: takes ownership
  T* release() { return p_.release(); }

 private:
  std::unique_ptr<T> p_;
};

}  // namespace mini
}  // namespace protobuf
}  // namespace google

#endif  // GOOGLE_PROTOBUF_MINI_RUNTIME_H_
