// Internal declarations shared by the translation units of libdvhip.so.
#ifndef DV_INTERNAL_H_
#define DV_INTERNAL_H_

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "dvhip.h"

namespace dv {

void set_error(const std::string& msg);
int fail(int status, const std::string& msg);

#define DV_HIP_CHECK(expr)                                                    \
  do {                                                                        \
    hipError_t _e = (expr);                                                   \
    if (_e != hipSuccess) {                                                   \
      return ::dv::fail(DV_ERR_HIP, std::string(#expr) + ": " +               \
                                        hipGetErrorString(_e));               \
    }                                                                         \
  } while (0)

// Grow-only device buffer.
struct DeviceBuffer {
  void* ptr = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes);
  void release();
};

// Optional per-launch timing with HIP events on the launch stream.
enum ProfileKind { kProfEncoder = 0, kProfConv = 1, kProfOther = 2, kProfKinds = 3 };
bool profiling_enabled();
void profile_begin(int kind, hipStream_t stream);
void profile_end(int kind, hipStream_t stream);

struct ProfileScope {
  int kind;
  hipStream_t stream;
  ProfileScope(int k, hipStream_t s) : kind(k), stream(s) {
    if (profiling_enabled()) profile_begin(kind, stream);
  }
  ~ProfileScope() {
    if (profiling_enabled()) profile_end(kind, stream);
  }
};

}  // namespace dv

#endif  // DV_INTERNAL_H_
