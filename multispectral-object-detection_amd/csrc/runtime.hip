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

extern "C" int cft_abi_version(void) { return 11; }   // 2: CFT_F16, dtype arguments of cft_bottleneck / cft_focus_conv, cft_to_nhwc; 3: w2_stages; 4: round 3 (probe exports removed, w2_stages required, permuted-row weights); 5: cft_clock_probe; 6: cft_gpt_upsample_add2, per-thread conv variant; 7: cft_conv2d_chain; 8: cft_conv2d_chain_ok takes ldx / ldy (the launcher's own validation); 9: cft_conv2d_chain_res, cft_linear_splitk, cft_layernorm_reduce; 10: cft_stem; 11: cft_stem / cft_stem_ok removed (probe build only), cft_set_conv_variant 96 / 961-964 / 97

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

// Shader-clock probe (measurement only, bench.py's sustained leg): one wave runs a dependent chain of v_fma_f32 for `spin_us`
// microseconds of the constant-rate wall clock (s_memrealtime) and reports the shader-clock ticks (s_memtime) and the FMAs it got
// done meanwhile.  Launched on a side stream next to the forward, it reads the clock the SIMDs actually run at under that load:
//   out[0] = shader-clock ticks, out[1] = wall-clock ticks, out[2] = dependent FMAs executed, out[3] = (keeps the chain live).
__global__ void __launch_bounds__(64) clock_probe_kernel(unsigned long long* out, unsigned long long spin_ticks) {
  const unsigned long long w0 = wall_clock64();
  const unsigned long long c0 = clock64();
  float a = 1.0f + threadIdx.x * 1e-7f;
  const float b = 1.0000001f;
  unsigned long long w1, n = 0;
  do {
#pragma unroll
    for (int i = 0; i < 256; ++i) a = __builtin_fmaf(a, b, 1e-9f);
    n += 256;
    w1 = wall_clock64();
  } while (w1 - w0 < spin_ticks);
  const unsigned long long c1 = clock64();
  if (threadIdx.x == 0) {
    out[0] = c1 - c0;
    out[1] = w1 - w0;
    out[2] = n;
    out[3] = (unsigned long long)__float_as_uint(a);
  }
}

extern "C" int cft_clock_probe(void* out4_u64, int spin_us, int* wall_khz, void* stream) {
  CFT_REQUIRE(out4_u64 != nullptr && wall_khz != nullptr && spin_us > 0 && spin_us <= 100000, "cft_clock_probe: bad argument");
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) {
    cft_set_error("cft_clock_probe: wall clock rate not available");
    return CFT_ENODEV;
  }
  *wall_khz = khz;
  const unsigned long long ticks = (unsigned long long)spin_us * (unsigned long long)khz / 1000ull;
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, as_stream(stream), (unsigned long long*)out4_u64, ticks);
  return cft_check_launch("clock_probe_kernel");
}
