// Shared host-side plumbing for libmkws_hip.so: error reporting across the C-ABI, HIP call checks.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/mkws.h"

namespace mkws {

inline char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

#define MKWS_HIP(expr)                                                                         \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      return ::mkws::fail(MKWS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                          __FILE__, __LINE__);                                                 \
  } while (0)

// Fails loudly when there is no usable gfx950 device: there is no CPU fallback in this library.
inline int require_device() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return fail(MKWS_ERR_NO_DEVICE, "no HIP device visible (libmkws_hip has no CPU fallback)");
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail(MKWS_ERR_NO_DEVICE, "hipGetDevice failed");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail(MKWS_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(MKWS_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", dev, prop.gcnArchName);
  return MKWS_OK;
}

}  // namespace mkws
