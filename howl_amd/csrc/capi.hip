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

// Side lanes are handed out per host thread and device but owned by one process-wide list, so that howl_shutdown can
// release all of them; a thread's cached slot is valid for the generation it was made in.
namespace {
std::mutex g_lane_mu;
std::vector<HowlSideLane*> g_lanes;
unsigned g_lane_generation = 1;
thread_local bool g_pending_error = false;
}  // namespace

HowlSideLane* howl_side_lane() {
    struct Slot {
        unsigned generation = 0;
        bool ok = false;
        HowlSideLane* lane = nullptr;
    };
    static thread_local Slot slots[16];
    if (getenv("HOWL_NO_SIDE_STREAM") != nullptr) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    Slot& s = slots[dev];
    std::lock_guard<std::mutex> lk(g_lane_mu);
    if (s.generation != g_lane_generation) {
        s.generation = g_lane_generation;
        HowlSideLane* l = new HowlSideLane();
        bool st = hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking) == hipSuccess;
        bool e1 = st && hipEventCreateWithFlags(&l->fork_ev, hipEventDisableTiming) == hipSuccess;
        bool e2 = e1 && hipEventCreateWithFlags(&l->join_ev, hipEventDisableTiming) == hipSuccess;
        s.ok = e2;
        if (s.ok) {
            s.lane = l;
            g_lanes.push_back(l);
        } else {
            (void)hipGetLastError();
            if (e1) hipEventDestroy(l->fork_ev);
            if (st) hipStreamDestroy(l->stream);
            delete l;
            s.lane = nullptr;
        }
    }
    return s.ok ? s.lane : nullptr;
}

// Raises a kernel's dynamic-LDS limit on the current device once per (host thread, device, size): `granted` is the caller's
// per-kernel table of 16 devices.  A refused raise is reported (howl_last_error names the kernel's request) and NOT cached, so
// the next call tries again; the entry point's HOWL_CHECK_LAUNCH returns HOWL_E_LAUNCH for it.
bool howl_raise_lds(const void* kernel, size_t lds, size_t* granted, const char* what) {
    int dev = 0;
    const bool indexed = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16;
    if (indexed && lds <= granted[dev]) return true;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        howl_set_error("%s: the runtime refused %zu bytes of dynamic LDS on device %d: %s", what, lds, dev, hipGetErrorString(e));
        g_pending_error = true;
        return false;
    }
    if (indexed) granted[dev] = lds;
    return true;
}

bool howl_take_pending_error() {
    const bool p = g_pending_error;
    g_pending_error = false;
    return p;
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
    {
        // the side lanes of every host thread (howl_side_lane): a later call on any thread makes a fresh one
        std::lock_guard<std::mutex> lk(g_lane_mu);
        for (HowlSideLane* l : g_lanes) {
            hipStreamSynchronize(l->stream);
            hipEventDestroy(l->fork_ev);
            hipEventDestroy(l->join_ev);
            hipStreamDestroy(l->stream);
            delete l;
        }
        g_lanes.clear();
        ++g_lane_generation;
    }
    (void)hipGetLastError();
    return HOWL_OK;
}

}  // extern "C"
