// TEST INFRASTRUCTURE (oracle/ref_build): absl ascii helpers.
#ifndef DVREF_ABSL_ASCII_H_
#define DVREF_ABSL_ASCII_H_
#include <cctype>
#include <string>
#include <string_view>
namespace absl {
inline bool ascii_isdigit(unsigned char c) { return std::isdigit(c) != 0; }
inline bool ascii_isalpha(unsigned char c) { return std::isalpha(c) != 0; }
inline bool ascii_isspace(unsigned char c) { return std::isspace(c) != 0; }
inline char ascii_toupper(unsigned char c) { return static_cast<char>(std::toupper(c)); }
inline char ascii_tolower(unsigned char c) { return static_cast<char>(std::tolower(c)); }
inline std::string AsciiStrToUpper(std::string_view s) { std::string o(s); for (char& c : o) c = ascii_toupper(static_cast<unsigned char>(c)); return o; }
inline std::string AsciiStrToLower(std::string_view s) { std::string o(s); for (char& c : o) c = ascii_tolower(static_cast<unsigned char>(c)); return o; }
inline void AsciiStrToUpper(std::string* s) { for (char& c : *s) c = ascii_toupper(static_cast<unsigned char>(c)); }
}  // namespace absl
#endif
