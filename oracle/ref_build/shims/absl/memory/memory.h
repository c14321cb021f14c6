// TEST INFRASTRUCTURE (oracle/ref_build): absl::make_unique / WrapUnique.
#ifndef DVREF_ABSL_MEMORY_H_
#define DVREF_ABSL_MEMORY_H_
#include <memory>
namespace absl {
using std::make_unique;
template <class T>
std::unique_ptr<T> WrapUnique(T* p) { return std::unique_ptr<T>(p); }
}  // namespace absl
#endif
