// (1 of 2: error state and the three informational entry points.)  Stand-ins for the entry points that live in .hip translation units, so that the HOST-ONLY sources of libdvhip.so
// can be linked into a sanitizer build (tools/host_asan/run.sh) and loaded by deepvariant_amd/_lib.py through
// DV_LIB_PATH.  Development tool: every device entry point answers DV_ERR_NO_DEVICE.
#include <string>

#include "dv_internal.h"

namespace dv {
static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }
int fail(int status, const std::string& msg) {
  g_error = msg;
  return status;
}
}  // namespace dv

extern "C" const char* dv_last_error(void) { return dv::g_error.c_str(); }
extern "C" int dv_abi_version(void) { return DV_ABI_VERSION; }
extern "C" int dv_device_count(void) { return 0; }
extern "C" int dv_host_asan_no_device(const char* name) { return dv::fail(DV_ERR_NO_DEVICE, std::string(name) + ": sanitizer build, host only"); }
