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
  int device = -1;   // the device `ptr` was allocated on (reserve() re-allocates after a device change)
  int reserve(size_t bytes);
  int reserve_on_current_device(size_t bytes);
  void release();
};

// Optional per-launch timing with HIP events on the launch stream.
enum ProfileKind { kProfEncoder = 0, kProfConv = 1, kProfOther = 2, kProfKinds = 3 };
// Callable from any number of host threads: a scope owns its event pair between begin and
// end, the shared lists behind them are guarded by a mutex (common.hip).
struct ProfileEvents {
  hipEvent_t a = nullptr, b = nullptr;
};
bool profiling_enabled();
ProfileEvents profile_begin(hipStream_t stream);
void profile_end(int kind, const ProfileEvents& ev, hipStream_t stream);

struct ProfileScope {
  int kind;
  hipStream_t stream;
  bool on;
  ProfileEvents ev;
  ProfileScope(int k, hipStream_t s) : kind(k), stream(s), on(profiling_enabled()) {
    if (on) ev = profile_begin(stream);
  }
  ~ProfileScope() {
    if (on) profile_end(kind, ev, stream);
  }
};

}  // namespace dv

#endif  // DV_INTERNAL_H_
