// TEST INFRASTRUCTURE (oracle/ref_build): absl::Status / StatusOr, the interface the reference's encoder uses.
#ifndef DVREF_ABSL_STATUS_H_
#define DVREF_ABSL_STATUS_H_
#include <optional>
#include <ostream>
#include <string>
#include <string_view>
#include <utility>
namespace absl {
enum class StatusCode { kOk = 0, kInvalidArgument = 3, kNotFound = 5, kFailedPrecondition = 9, kOutOfRange = 11, kInternal = 13, kUnknown = 2 };
class Status {
 public:
  Status() = default;
  Status(StatusCode c, std::string_view m) : code_(c), msg_(m) {}
  bool ok() const { return code_ == StatusCode::kOk; }
  StatusCode code() const { return code_; }
  std::string_view message() const { return msg_; }
  std::string ToString() const { return ok() ? "OK" : msg_; }
 private:
  StatusCode code_ = StatusCode::kOk;
  std::string msg_;
};
inline std::ostream& operator<<(std::ostream& o, const Status& s) { return o << s.ToString(); }
inline Status OkStatus() { return Status(); }
inline Status InvalidArgumentError(std::string_view m) { return Status(StatusCode::kInvalidArgument, m); }
inline Status NotFoundError(std::string_view m) { return Status(StatusCode::kNotFound, m); }
inline Status FailedPreconditionError(std::string_view m) { return Status(StatusCode::kFailedPrecondition, m); }
inline Status OutOfRangeError(std::string_view m) { return Status(StatusCode::kOutOfRange, m); }
inline Status InternalError(std::string_view m) { return Status(StatusCode::kInternal, m); }
inline Status UnknownError(std::string_view m) { return Status(StatusCode::kUnknown, m); }
template <class T>
class StatusOr {
 public:
  StatusOr(const T& v) : v_(v) {}
  StatusOr(T&& v) : v_(std::move(v)) {}
  StatusOr(const Status& s) : s_(s) {}
  bool ok() const { return v_.has_value(); }
  const Status& status() const { return s_; }
  const T& value() const& { return *v_; }
  T& value() & { return *v_; }
  T&& value() && { return std::move(*v_); }
  const T& operator*() const& { return *v_; }
  T& operator*() & { return *v_; }
  const T* operator->() const { return &*v_; }
  T* operator->() { return &*v_; }
 private:
  std::optional<T> v_;
  Status s_;
};
}  // namespace absl
#endif
