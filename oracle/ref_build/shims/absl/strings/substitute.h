// TEST INFRASTRUCTURE (oracle/ref_build): absl::Substitute ("$0 ... $9" positional arguments).
#ifndef DVREF_ABSL_SUBSTITUTE_H_
#define DVREF_ABSL_SUBSTITUTE_H_
#include <sstream>
#include <string>
#include <string_view>
#include <vector>
namespace absl {
template <class... A>
std::string Substitute(std::string_view format, const A&... a) {
  std::vector<std::string> args;
  auto add = [&](const auto& v) {
    std::ostringstream s;
    s << v;
    args.push_back(s.str());
  };
  (add(a), ...);
  std::string out;
  for (size_t i = 0; i < format.size(); ++i) {
    if (format[i] == '$' && i + 1 < format.size()) {
      const char c = format[i + 1];
      if (c >= '0' && c <= '9' && static_cast<size_t>(c - '0') < args.size()) {
        out += args[static_cast<size_t>(c - '0')];
        ++i;
        continue;
      }
      if (c == '$') {
        out += '$';
        ++i;
        continue;
      }
    }
    out += format[i];
  }
  return out;
}
}  // namespace absl
#endif
