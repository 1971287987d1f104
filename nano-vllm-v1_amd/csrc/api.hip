// Library-level entry points: version, thread-local error string, device check.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include "common.h"

static thread_local char g_err[512] = "";

void nvh_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* nvh_last_error(void) { return g_err; }

// ---- routing / measurement switches (common.h)
namespace {
const char* const kSwitchNames[SW_COUNT] = {
#define NVH_SW_NAME(n) "NVH_" #n,
    NVH_SWITCHES(NVH_SW_NAME)
#undef NVH_SW_NAME
};
std::atomic<int64_t> g_switch[SW_COUNT];
std::once_flag g_switch_once;
void load_switches() {
    for (int i = 0; i < SW_COUNT; ++i) g_switch[i].store(-1, std::memory_order_relaxed);
    const char* tuning = getenv("NVH_TUNING");
    if (!tuning || strcmp(tuning, "1") != 0) return;
    for (int i = 0; i < SW_COUNT; ++i) {
        const char* v = getenv(kSwitchNames[i]);
        if (!v) continue;
        char* end = nullptr;
        const long long n = strtoll(v, &end, 10);
        g_switch[i].store((end != v && n >= 0) ? (int64_t)n : 1, std::memory_order_relaxed);
    }
}
}  // namespace

int64_t nvh_switch(NvhSwitch s) {
    std::call_once(g_switch_once, load_switches);
    return g_switch[s].load(std::memory_order_relaxed);
}

extern "C" int32_t nvh_set_switch(const char* name, int64_t value) {
    NVH_REQUIRE(name, NVH_EINVAL, "nvh_set_switch: null name");
    std::call_once(g_switch_once, load_switches);
    for (int i = 0; i < SW_COUNT; ++i) {
        if (strcmp(name, kSwitchNames[i]) == 0 || strcmp(name, kSwitchNames[i] + 4) == 0) {
            g_switch[i].store(value < 0 ? -1 : value, std::memory_order_relaxed);
            return NVH_OK;
        }
    }
    nvh_set_error("nvh_set_switch: unknown switch %s", name);
    return NVH_EINVAL;
}

extern "C" int64_t nvh_get_switch(const char* name) {
    if (!name) return -2;
    for (int i = 0; i < SW_COUNT; ++i)
        if (strcmp(name, kSwitchNames[i]) == 0 || strcmp(name, kSwitchNames[i] + 4) == 0) return nvh_switch((NvhSwitch)i);
    return -2;
}

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
