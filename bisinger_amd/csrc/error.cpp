// Thread-local error string + ABI/version probes of libbisinger_hip.
#include <stdarg.h>
#include <string.h>

#include "bsg_common.h"

namespace bsg {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace bsg

extern "C" int bsg_abi_version(void) { return BSG_ABI_VERSION; }
extern "C" const char* bsg_last_error(void) { return bsg::g_err; }
extern "C" const char* bsg_device_arch(void) {
  static thread_local char arch[256];
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return nullptr;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) return nullptr;
  strncpy(arch, p.gcnArchName, sizeof(arch) - 1);
  return arch;
}
