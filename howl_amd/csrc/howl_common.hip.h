// Shared device/host helpers for the gfx950 kernels behind include/howl_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
// explicit LDS (address space 3) pointer: keeps pointer arithmetic on tile addresses as ds_read with immediate
// offsets (a generic float* that the optimiser cannot trace back to LDS degrades to flat_load)
typedef __attribute__((address_space(3))) float lds_f32;

// Makes a VGPR value opaque to the optimiser at this point: what is derived from it afterwards (an LDS address, a byte
// offset) is recomputed where it is used -- one or two VALU instructions -- instead of being hoisted out of the enclosing
// loop into registers of its own (the 3x3 convolution kernels sit at the 168-VGPR line of three waves per SIMD; hoisted
// per-slot addresses were spilled to scratch and reloaded in front of every use).  No code is emitted.
#if defined(HIPEMU)
#define HOWL_OPAQUE_V(x) ((void)(x))
#else
#define HOWL_OPAQUE_V(x) asm volatile("" : "+v"(x))
#endif
// The same for an LDS cursor (a 32-bit address in a VGPR): behind this point the pointer is a plain register value, so every
// access off it is "register + immediate" -- otherwise the optimiser keeps cursors as (dynamic-LDS symbol + constant + lane
// part) and re-adds the constant in front of each ds_read when it exceeds the 16-bit offset field (the 3x3 forward K loop
// carried 43 v_add_u32 per 45 MFMAs, and vector instructions do not overlap the matrix pipe on this part).
#define HOWL_OPAQUE_LDS(p) HOWL_OPAQUE_V(p)

#define HOWL_OK 0
#define HOWL_E_ARG (-1)       // bad argument (shape / null pointer / unsupported size)
#define HOWL_E_LAUNCH (-2)    // hipGetLastError() after a launch reported a failure
#define HOWL_E_WORKSPACE (-3) // caller-provided workspace too small

void howl_set_error(const char* fmt, ...);
int howl_num_cus();
bool howl_prof_begin(const char* tag, hipStream_t stream, size_t* slot, double work);
void howl_prof_end(size_t slot, hipStream_t stream);

// A second HIP queue of the library's own for launches that do not depend on each other INSIDE one entry point (round 5: the
// head's weight gradient next to the backward recurrence, which occupies half of the CUs).  One lane per host thread and device,
// created on first use and kept; fork = the lane waits for everything queued on the caller's stream so far, join = the caller's
// stream waits for everything queued on the lane -- stream-ordered on both sides, the host never blocks, and when the entry point
// returns all of its work is ordered before whatever the caller queues next.  nullptr: no lane (HOWL_NO_SIDE_STREAM set, or the
// runtime refused one): callers then launch on the caller's stream.
struct HowlSideLane {
    hipStream_t stream;
    hipEvent_t fork_ev, join_ev;
};
HowlSideLane* howl_side_lane();
// dynamic-LDS limit of `kernel` on the current device raised to `lds` bytes (cached per host thread and device in the caller's
// table of 16); false = refused: the error text is set and the entry point's HOWL_CHECK_LAUNCH will return HOWL_E_LAUNCH
bool howl_raise_lds(const void* kernel, size_t lds, size_t* granted, const char* what);
bool howl_take_pending_error();
inline void howl_lane_fork(HowlSideLane* l, hipStream_t main) {
    hipEventRecord(l->fork_ev, main);
    hipStreamWaitEvent(l->stream, l->fork_ev, 0);
}
inline void howl_lane_join(HowlSideLane* l, hipStream_t main) {
    hipEventRecord(l->join_ev, l->stream);
    hipStreamWaitEvent(main, l->join_ev, 0);
}

// brackets the launches in its scope with HIP events when howl_profile_enable(1) is active (no-op otherwise); `work` =
// the algorithmic FLOPs (or bytes) of those launches, summed per tag by howl_profile_read_work
struct HowlProfScope {
    size_t slot = 0;
    bool on;
    hipStream_t stream;
    HowlProfScope(const char* tag, hipStream_t s, double work = 0.0) : stream(s) { on = howl_prof_begin(tag, s, &slot, work); }
    ~HowlProfScope() {
        if (on) howl_prof_end(slot, stream);
    }
};

// One element of torch.optim.AdamW (decoupled weight decay; pretrain_gsc.py:93,133, train.py:256,302): the arithmetic of
// howl_adamw_step's kernel, shared with the slab fold that applies the step to the gradients it has just summed.
struct HowlAdamWCoef {
    float lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale;
};
__device__ __forceinline__ void howl_adamw_element(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, size_t i, float g,
                                                   const HowlAdamWCoef& c) {
    // no multiply-add contraction here: the same element must come out bit-identical whichever kernel applies the step (the
    // optimiser's own launch, a slab fold, res8's last fold launch) -- contraction is a per-call-site decision of the compiler
#pragma clang fp contract(off)
    const float gi = g * c.gscale;
    float pi = p[i] * (1.0f - c.lr * c.wd);
    const float mi = m[i] + (gi - m[i]) * (1.0f - c.b1);
    const float vi = c.b2 * v[i] + (1.0f - c.b2) * gi * gi;
    const float denom = sqrtf(vi) / c.bc2_sqrt + c.eps;
    pi -= (c.lr / c.bc1) * (mi / denom);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
}

#define HOWL_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            howl_set_error(__VA_ARGS__);        \
            return HOWL_E_ARG;                  \
        }                                       \
    } while (0)

#define HOWL_CHECK_LAUNCH(name)                                                     \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (howl_take_pending_error()) return HOWL_E_LAUNCH; /* text already set */  \
        if (e_ != hipSuccess) {                                                     \
            howl_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
            return HOWL_E_LAUNCH;                                                   \
        }                                                                           \
    } while (0)

// Lanes of ONE wavefront exchange data through LDS without a workgroup barrier: a wave's DS operations
// are issued in order, so a wave-scope release/acquire pair (which lowers to s_waitcnt lgkmcnt(0) only)
// plus the compiler-level wave barrier is sufficient on gfx950.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float wave_sum(float v) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
// The same sum with the first four butterfly stages inside a row of 16 lanes as DPP moves (quad_perm [1,0,3,2], quad_perm
// [2,3,0,1], row_half_mirror, row_mirror: no LDS-pipe traffic) and only the two cross-row stages as ds_bpermute -- for kernels
// that fold MANY accumulators per wave at their end (conv0's weight gradient: 27 per wave, 162 -> 54 ds_bpermute).  A different
// association order than wave_sum: not interchangeable where bits are compared.
__device__ __forceinline__ float wave_sum_rows(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
