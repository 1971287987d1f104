// Library-level entry points: version, thread-local error string, device check.
#include <stdarg.h>
#include <string.h>
#include "common.h"

static thread_local char g_err[512] = "";

void nvh_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* nvh_last_error(void) { return g_err; }

extern "C" int32_t nvh_version(void) { return 100; }  // 0.1.0

extern "C" int32_t nvh_device_check(int32_t device_index) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device_index);
    NVH_REQUIRE(e == hipSuccess, NVH_ENODEV, "nvh_device_check: hipGetDeviceProperties(%d): %s", device_index, hipGetErrorString(e));
    NVH_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, NVH_ENODEV,
                "nvh_device_check: device %d is %s; this library contains gfx950 code only", device_index, prop.gcnArchName);
    // set-up for nvh_attn_decode_noalloc: resolve this GPU's copy of its built-in array now when the checked device is
    // the current one (otherwise the entry point does it on its first call there)
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == device_index) {
        const int32_t rc = nvh_noalloc_scratch(nullptr);
        return rc != NVH_OK ? rc : nvh_decode_arrivals(nullptr);
    }
    return NVH_OK;
}
