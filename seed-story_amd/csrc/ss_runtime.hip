// Library-wide host runtime: thread-local error string, tuning knobs, device probe.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>

#include "ss_common.h"

namespace ss {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return SS_OK;
    set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
    return SS_EHIP;
}

static std::mutex g_tune_mu;
static std::map<std::string, int>& tune_map() {
    static std::map<std::string, int> m;
    return m;
}

int tuning_get(const char* key, int dflt) {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    auto& m = tune_map();
    auto it = m.find(key);
    return it == m.end() ? dflt : it->second;
}

void tuning_set(const char* key, int value) {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    tune_map()[key] = value;
}

}  // namespace ss

extern "C" {

const char* ss_last_error(void) { return ss::g_err; }
int ss_abi_version(void) { return SS_ABI_VERSION; }

int ss_set_tuning(const char* key, int value) {
    if (!key) return SS_EINVAL;
    std::lock_guard<std::mutex> lk(ss::g_tune_mu);
    ss::tune_map()[key] = value;
    return SS_OK;
}
int ss_get_tuning(const char* key, int dflt) { return key ? ss::tuning_get(key, dflt) : dflt; }

int ss_device_info(int32_t out[4]) {
    int dev = 0;
    SS_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    SS_HIP(hipGetDeviceProperties(&p, dev));
    out[0] = p.multiProcessorCount;
    out[1] = strstr(p.gcnArchName, "gfx950") != nullptr;
    out[2] = (int32_t)(p.totalGlobalMem >> 20);
    out[3] = p.warpSize;
    return SS_OK;
}

}  // extern "C"
