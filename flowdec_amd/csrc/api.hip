// api.hip -- error plumbing and device queries of the C ABI.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

int fd_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* fd_last_error(void) { return g_err; }
extern "C" int fd_version(void) { return 100; }

extern "C" int fd_device_info(int* out4) {
  FD_REQUIRE(out4, "fd_device_info: null pointer");
  int dev = 0;
  FD_HIP(hipGetDevice(&dev));
  hipDeviceProp_t p;
  FD_HIP(hipGetDeviceProperties(&p, dev));
  out4[0] = p.multiProcessorCount;
  out4[1] = p.clockRate;
  out4[2] = p.warpSize;
  int arch = 0;
  for (const char* c = p.gcnArchName; *c && *c != ':'; ++c)
    if (*c >= '0' && *c <= '9') arch = arch * 10 + (*c - '0');
  out4[3] = arch;
  return FD_OK;
}
