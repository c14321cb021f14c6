// TEST INFRASTRUCTURE (oracle/ref_build): the slice of abseil the reference's encoder sources use, on the standard library.
#ifndef DVREF_ABSL_STRING_VIEW_H_
#define DVREF_ABSL_STRING_VIEW_H_
#include <string_view>
namespace absl {
using string_view = std::string_view;
}
#endif
