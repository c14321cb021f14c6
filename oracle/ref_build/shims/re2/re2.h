// TEST INFRASTRUCTURE (oracle/ref_build): the two RE2 calls the reference's realigner makes (a pattern object and
// RE2::Consume with captured arguments: fast_pass_aligner.cc:347-350, 623-630) on std::regex.
#ifndef DVREF_RE2_SHIM_H_
#define DVREF_RE2_SHIM_H_
#include <cstdlib>
#include <regex>
#include <string>
#include <string_view>
namespace re2 {
class RE2 {
 public:
  RE2(const char* pattern) : re_(pattern) {}
  RE2(const std::string& pattern) : re_(pattern) {}
  // Matches at the START of *input; on success stores the captures and removes the match from *input.
  template <class... A>
  static bool Consume(std::string_view* input, const RE2& re, A*... args) {
    std::cmatch m;
    if (!std::regex_search(input->data(), input->data() + input->size(), m, re.re_, std::regex_constants::match_continuous)) {
      return false;
    }
    size_t k = 1;
    bool ok = true;
    ((ok = ok && Store(m, k++, args)), ...);
    if (!ok) return false;
    input->remove_prefix(static_cast<size_t>(m.length(0)));
    return true;
  }
  // Replaces the first match of `re` in *str with `rewrite` (no backreferences used by the callers).
  static bool Replace(std::string* str, const RE2& re, std::string_view rewrite) {
    std::smatch m;
    if (!std::regex_search(*str, m, re.re_)) return false;
    str->replace(static_cast<size_t>(m.position(0)), static_cast<size_t>(m.length(0)), rewrite);
    return true;
  }
 private:
  static bool Store(const std::cmatch& m, size_t k, int* out) {
    if (k >= m.size()) return false;
    *out = std::atoi(m[k].str().c_str());
    return true;
  }
  static bool Store(const std::cmatch& m, size_t k, std::string* out) {
    if (k >= m.size()) return false;
    *out = m[k].str();
    return true;
  }
  std::regex re_;
};
}  // namespace re2
using re2::RE2;
#endif
