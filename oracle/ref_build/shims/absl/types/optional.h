// TEST INFRASTRUCTURE (oracle/ref_build): absl::optional = std::optional.
#ifndef DVREF_ABSL_OPTIONAL_H_
#define DVREF_ABSL_OPTIONAL_H_
#include <optional>
namespace absl {
template <class T>
using optional = std::optional<T>;
using std::nullopt;
using std::make_optional;
}  // namespace absl
#endif
