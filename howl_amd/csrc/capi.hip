// Library-level entry points of libhowl_hip.so: version, thread-local error text, device query cache.
#include <stdarg.h>

#include "howl_common.hip.h"
#include "../../include/howl_hip.h"

#include <stdlib.h>

#include <mutex>
#include <string>
#include <vector>

namespace {
thread_local char g_err[512] = "";

// optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg)
struct ProfRec {
    std::string tag;
    hipEvent_t start, stop;
    double work;   // algorithmic FLOPs or bytes of the bracketed launches (whatever the call site counts)
};
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfRec> g_prof;

}  // namespace

bool howl_prof_begin(const char* tag, hipStream_t stream, size_t* slot, double work) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_on) return false;
    ProfRec r;
    r.tag = tag;
    r.work = work;
    if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return false;
    hipEventRecord(r.start, stream);
    g_prof.push_back(r);
    *slot = g_prof.size() - 1;
    return true;
}

void howl_prof_end(size_t slot, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (slot < g_prof.size()) hipEventRecord(g_prof[slot].stop, stream);
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

HowlSideLane* howl_side_lane() {
    struct Slot {
        int dev = -1;
        bool tried = false, ok = false;
        HowlSideLane lane;
    };
    static thread_local Slot slots[16];
    if (getenv("HOWL_NO_SIDE_STREAM") != nullptr) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    Slot& s = slots[dev];
    if (!s.tried) {
        s.tried = true;
        s.dev = dev;
        s.ok = hipStreamCreateWithFlags(&s.lane.stream, hipStreamNonBlocking) == hipSuccess &&
               hipEventCreateWithFlags(&s.lane.fork_ev, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&s.lane.join_ev, hipEventDisableTiming) == hipSuccess;
        if (!s.ok) (void)hipGetLastError();
    }
    return s.ok ? &s.lane : nullptr;
}

extern "C" {

int howl_version(int* major, int* minor) {
    if (major) *major = 0;
    if (minor) *minor = 1;
    return HOWL_OK;
}

const char* howl_last_error(void) { return g_err; }

int howl_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return HOWL_OK;
}

int howl_profile_read_work(const char* tag, double* total_ms, int* count, double* work, int reset) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double tot = 0.0, wk = 0.0;
    int n = 0;
    for (auto& r : g_prof) {
        if (tag != nullptr && r.tag != tag) continue;
        if (hipEventSynchronize(r.stop) != hipSuccess) continue;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
            tot += (double)ms;
            wk += r.work;
            ++n;
        }
    }
    if (total_ms) *total_ms = tot;
    if (count) *count = n;
    if (work) *work = wk;
    if (reset) {
        for (auto& r : g_prof) {
            hipEventDestroy(r.start);
            hipEventDestroy(r.stop);
        }
        g_prof.clear();
    }
    return HOWL_OK;
}

int howl_profile_read(const char* tag, double* total_ms, int* count, int reset) {
    return howl_profile_read_work(tag, total_ms, count, nullptr, reset);
}

int howl_shutdown(void) {
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof_on = false;
        for (auto& r : g_prof) {
            hipEventDestroy(r.start);
            hipEventDestroy(r.stop);
        }
        g_prof.clear();
    }
    return HOWL_OK;
}

}  // extern "C"
