// Library-level plumbing of libcft_hip.so: error string, launch check, device probe.
#include "cft_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void cft_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}

int cft_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    char buf[512];
    snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    cft_set_error(buf);
    return CFT_ELAUNCH;
  }
  return CFT_OK;
}

extern "C" int cft_abi_version(void) { return 4; }   // 2: CFT_F16, dtype arguments of cft_bottleneck / cft_focus_conv, cft_to_nhwc; 3: w2_stages; 4: round 3 (probe exports removed, w2_stages required, permuted-row weights)

extern "C" const char* cft_last_error(void) { return g_err; }

extern "C" int cft_device_check(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { cft_set_error("cft_device_check: no HIP device"); return CFT_ENODEV; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { cft_set_error("cft_device_check: hipGetDeviceProperties failed"); return CFT_ENODEV; }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    char buf[256];
    snprintf(buf, sizeof(buf), "cft_device_check: kernels are built for gfx950 only, device is %s", prop.gcnArchName);
    cft_set_error(buf);
    return CFT_ENODEV;
  }
  return CFT_OK;
}
