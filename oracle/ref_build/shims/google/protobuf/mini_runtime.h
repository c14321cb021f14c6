// mini_runtime.h -- the container types behind the structs oracle/ref_build/mini_protoc.py generates.
// TEST INFRASTRUCTURE (oracle/): stands in for libprotobuf's RepeatedField / RepeatedPtrField / Map so that the
// reference's sources compile unmodified.  Only the interface those sources use; same semantics
// (int sizes, `at` on a missing key is fatal, default instances for unset message fields).
#ifndef GOOGLE_PROTOBUF_MINI_RUNTIME_H_
#define GOOGLE_PROTOBUF_MINI_RUNTIME_H_

#include <stdexcept>
#include <map>
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

template <class T>
class RepeatedPtrField : public std::vector<T> {
 public:
  using std::vector<T>::vector;
  int size() const { return static_cast<int>(std::vector<T>::size()); }
  const T& Get(int i) const { return (*this)[i]; }
  T* Mutable(int i) { return &(*this)[i]; }
  T* Add() {
    this->emplace_back();
    return &this->back();
  }
  void Add(T&& v) { this->push_back(std::move(v)); }
  void Add(const T& v) { this->push_back(v); }
  void Clear() { this->clear(); }
  void Reserve(int n) { this->reserve(n); }
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

 private:
  std::unique_ptr<T> p_;
};

}  // namespace mini
}  // namespace protobuf
}  // namespace google

#endif  // GOOGLE_PROTOBUF_MINI_RUNTIME_H_
