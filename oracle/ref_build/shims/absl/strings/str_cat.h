// TEST INFRASTRUCTURE (oracle/ref_build): absl::StrCat / StrAppend on ostringstream.
#ifndef DVREF_ABSL_STR_CAT_H_
#define DVREF_ABSL_STR_CAT_H_
#include <sstream>
#include <string>
namespace absl {
template <class... A>
std::string StrCat(const A&... a) {
  std::ostringstream s;
  (s << ... << a);
  return s.str();
}
template <class... A>
void StrAppend(std::string* dst, const A&... a) { dst->append(StrCat(a...)); }
}  // namespace absl
#endif
