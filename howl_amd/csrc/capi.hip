// Library-level entry points of libhowl_hip.so: version, thread-local error text, device query cache.
#include <stdarg.h>

#include "howl_common.hip.h"
#include "../../include/howl_hip.h"

namespace {
thread_local char g_err[512] = "";
}

void howl_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int howl_num_cus() {
    static int cached = 0;  // benign race: every thread computes the same value
    if (cached == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
            prop.multiProcessorCount > 0)
            cached = prop.multiProcessorCount;
        else
            cached = 256;
    }
    return cached;
}

extern "C" {

int howl_version(int* major, int* minor) {
    if (major) *major = 0;
    if (minor) *minor = 1;
    return HOWL_OK;
}

const char* howl_last_error(void) { return g_err; }

}  // extern "C"
