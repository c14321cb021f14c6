// TEST INFRASTRUCTURE (oracle/ref_build): absl::StartsWith / EndsWith / StrContains.
#ifndef DVREF_ABSL_MATCH_H_
#define DVREF_ABSL_MATCH_H_
#include <string_view>
namespace absl {
inline bool StartsWith(std::string_view s, std::string_view p) { return s.size() >= p.size() && s.compare(0, p.size(), p) == 0; }
inline bool EndsWith(std::string_view s, std::string_view p) { return s.size() >= p.size() && s.compare(s.size() - p.size(), p.size(), p) == 0; }
inline bool StrContains(std::string_view s, std::string_view p) { return s.find(p) != std::string_view::npos; }
inline bool StrContains(std::string_view s, char c) { return s.find(c) != std::string_view::npos; }
}  // namespace absl
#endif
