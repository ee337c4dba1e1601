// rn_common.hip -- error reporting and library-level entry points of libradnerf_hip.so.
#include "rn_common.h"

#include <stdarg.h>
#include <stdio.h>

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

}  // namespace rn

extern "C" {

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
