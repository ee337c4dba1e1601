// rn_common.hip -- error reporting and library-level entry points of libradnerf_hip.so.
#include "rn_common.h"

#include <stdarg.h>
#include <stdio.h>

#include <utility>
#include <vector>

namespace rn {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: HIP error %d (%s)", what, (int)e, hipGetErrorString(e));
        return RN_ERR_LAUNCH;
    }
    return RN_OK;
}

// ---- optional HIP-event timing of one named kernel family (used by bench.py for the roofline object) ----
static bool g_prof_on = false, g_prof_paused = false;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_pool;
static size_t g_prof_used = 0;

bool prof_enabled() { return g_prof_on; }

// Events for the next timed launch (both null when timing is off).  They are attached to the kernel's own dispatch packet
// with hipExtLaunchKernelGGL: a pair of hipEventRecord calls around the launch put two barrier packets on the queue and
// cost ~5.7 us of idle GPU each (11 us per loop iteration, 8 % of a frame, seen in the rocprofv3 kernel trace).
void prof_pair(hipEvent_t *start, hipEvent_t *stop) {
    *start = *stop = nullptr;
    if (!g_prof_on || g_prof_paused) return;
    if (g_prof_used == g_prof_pool.size()) {
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        g_prof_pool.emplace_back(a, b);
    }
    *start = g_prof_pool[g_prof_used].first;
    *stop = g_prof_pool[g_prof_used].second;
    g_prof_used++;
}

}  // namespace rn

extern "C" {

int rn_prof_enable(int on) {
    rn::g_prof_on = on != 0;
    rn::g_prof_paused = false;
    rn::g_prof_used = 0;
    return RN_OK;
}

int rn_prof_pause(int paused) {
    rn::g_prof_paused = paused != 0;
    return RN_OK;
}

int rn_prof_collect(uint32_t *launches, float *total_ms) {
    float total = 0.0f;
    for (size_t i = 0; i < rn::g_prof_used; i++) {
        if (hipEventSynchronize(rn::g_prof_pool[i].second) != hipSuccess) return RN_ERR_LAUNCH;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, rn::g_prof_pool[i].first, rn::g_prof_pool[i].second) != hipSuccess) return RN_ERR_LAUNCH;
        total += ms;
    }
    if (launches) *launches = (uint32_t)rn::g_prof_used;
    if (total_ms) *total_ms = total;
    return RN_OK;
}

int rn_prof_durations(float *out_ms, uint32_t capacity) {
    const size_t n = rn::g_prof_used < capacity ? rn::g_prof_used : capacity;
    for (size_t i = 0; i < n; i++) {
        if (hipEventSynchronize(rn::g_prof_pool[i].second) != hipSuccess) return RN_ERR_LAUNCH;
        if (hipEventElapsedTime(&out_ms[i], rn::g_prof_pool[i].first, rn::g_prof_pool[i].second) != hipSuccess) return RN_ERR_LAUNCH;
    }
    return (int)n;
}

const char *rn_last_error(void) { return rn::g_err; }

int rn_version(void) { return 100; }  // 0.1.0

int rn_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        rn::set_error("no HIP device visible");
        return RN_ERR_NO_DEVICE;
    }
    return n;
}

}  // extern "C"
