// hipemu -- TEST INFRASTRUCTURE ONLY.  A single-threaded fiber emulator of the HIP execution model
// (workgroups, 64-lane wavefronts, LDS, __syncthreads, cross-lane ops, f32 MFMA) that lets the kernels
// under howl_amd/csrc be compiled UNMODIFIED with g++ and run on host memory for tiny shapes, so that
// indexing / fragment-layout / bounds mistakes are caught (with guard pages) in the CPU test-suite before a
// GPU run.  It is not a fallback: nothing under howl_amd/ knows about it and the product library is built
// by hipcc for gfx950 only.  MFMA semantics follow /opt/skills/guides/cdna_hip_programming.md section 3
// (k-ordered fmaf chain; A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D col=l&15,row=(l>>4)*4+reg for 16x16x4).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define HIPEMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__ __restrict

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };
extern uint3_emu threadIdx, blockIdx;
extern dim3 blockDim, gridDim;
extern char* hipemu_dynamic_lds;   // extern __shared__ arrays resolve to this via HOWL_DYNAMIC_LDS
static constexpr int warpSize = 64;

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }

typedef struct ihipStream_t* hipStream_t;
typedef int hipError_t;
static constexpr hipError_t hipSuccess = 0;
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
enum hipMemcpyKind { hipMemcpyDeviceToDevice = 3, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 };
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
    memcpy(d, s, n); return hipSuccess;
}
typedef int hipEvent_t;
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = 0; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
enum { hipEventDisableTiming = 2, hipStreamNonBlocking = 1 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = 0; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }  // the emulator runs launches synchronously
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
struct hipDeviceProp_t { int multiProcessorCount; };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
// 4 "CUs" by default (grids stay tiny); HIPEMU_CUS=n (read when the library is loaded: howl_num_cus caches) lets a test
// process reach the launch geometries that need more of them
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    const char* e = getenv("HIPEMU_CUS");
    p->multiProcessorCount = (e && atoi(e) > 0) ? atoi(e) : 4;
    return hipSuccess;
}
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
static constexpr int hipFuncAttributeMaxDynamicSharedMemorySize = 8;

// ---- scheduler interface --------------------------------------------------------------------------------
void hipemu_launch(const std::function<void()>& body, dim3 grid, dim3 block, size_t lds_bytes);
void hipemu_syncthreads();
// all 64 lanes of a wave deposit `n` 32-bit words; returns pointer to the wave's [64][16] word table
const uint32_t* hipemu_wave_exchange(const uint32_t* words, int n);
int hipemu_lane();

#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(hipemu_dynamic_lds);
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    hipemu_launch([=]() { kernel(__VA_ARGS__); }, dim3(grid), dim3(block), (size_t)(lds))

static inline void __syncthreads() { hipemu_syncthreads(); }
static inline void __builtin_amdgcn_s_barrier() { hipemu_syncthreads(); }
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline void __threadfence() {}
// device-scope relaxed atomics / wait counters: workgroups run one after another here, plain accesses are equivalent
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_store(p, v, order, scope) ((void)(*(p) = (v)))
#define __hip_atomic_load(p, order, scope) (*(p))
template <class T> static inline T hipemu_fetch_add(T* p, T v) { T o = *p; *p = o + v; return o; }
#define __hip_atomic_fetch_add(p, v, order, scope) hipemu_fetch_add((p), (v))
static inline void __builtin_amdgcn_s_waitcnt(int) {}
static inline void __builtin_amdgcn_s_sleep(int) {}

template <class T> static inline uint32_t hipemu_bits(T v) { uint32_t u; static_assert(sizeof(T) == 4, ""); memcpy(&u, &v, 4); return u; }
template <class T> static inline T hipemu_from(uint32_t u) { T v; memcpy(&v, &u, 4); return v; }

template <class T> static inline T __shfl(T v, int src, int width = 64) {
    uint32_t w = hipemu_bits(v);
    const uint32_t* t = hipemu_wave_exchange(&w, 1);
    int lane = hipemu_lane();
    int base = lane & ~(width - 1);
    return hipemu_from<T>(t[(base + (src & (width - 1))) * 16]);
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) { return __shfl(v, (hipemu_lane() ^ mask), 64); }
template <class T> static inline T __shfl_down(T v, int d, int width = 64) {
    int lane = hipemu_lane();
    int src = lane + d;
    if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return __shfl(v, src, 64);
}
static inline double __shfl_xor(double v, int mask) {
    uint32_t w[2]; memcpy(w, &v, 8);
    const uint32_t* t = hipemu_wave_exchange(w, 2);
    int src = hipemu_lane() ^ mask;
    uint32_t r[2] = {t[src * 16], t[src * 16 + 1]};
    double out; memcpy(&out, r, 8); return out;
}
static inline unsigned long long __ballot(int pred) {
    uint32_t w = pred ? 1u : 0u;
    const uint32_t* t = hipemu_wave_exchange(&w, 1);
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) if (t[l * 16]) m |= 1ull << l;
    return m;
}
static inline int __builtin_amdgcn_readfirstlane(int v) {
    uint32_t w = (uint32_t)v;
    const uint32_t* t = hipemu_wave_exchange(&w, 1);
    return (int)t[0];
}


// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[row=(l>>4)*4+r][col=l&15]; k-ordered fmaf chain
template <class V4> static inline V4 hipemu_mfma16(float a, float b, V4 c) {
    uint32_t w[2] = {hipemu_bits(a), hipemu_bits(b)};
    const uint32_t* t = hipemu_wave_exchange(w, 2);
    int l = hipemu_lane();
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            float av = hipemu_from<float>(t[(k * 16 + row) * 16 + 0]);
            float bv = hipemu_from<float>(t[(k * 16 + col) * 16 + 1]);
            acc = fmaf(av, bv, acc);
        }
        c[r] = acc;
    }
    return c;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) hipemu_mfma16((a), (b), (c))

// v_mfma_f32_4x4x1_16b_f32: 16 independent blocks j = l>>2; A_j[i] from lane 4j+i, B_j[c] from lane 4j+c; lane 4j+c holds
// D_j[r = 0..3][c] in its four result registers (layout probed on gfx950: tools/mfma4x4_probe.hip)
// cbsz / abid: block abid of every group of 2^cbsz blocks supplies A to the whole group (CDNA3 ISA 7.1.4; probed on gfx950)
template <class V4> static inline V4 hipemu_mfma4(float a, float b, V4 c, int cbsz = 0, int abid = 0) {
    uint32_t w[2] = {hipemu_bits(a), hipemu_bits(b)};
    const uint32_t* t = hipemu_wave_exchange(w, 2);
    const int l = hipemu_lane(), blk = l >> 2;
    const int ablk = (blk & ~((1 << cbsz) - 1)) | (cbsz ? abid : 0);
    const float bv = hipemu_from<float>(t[l * 16 + 1]);
    for (int r = 0; r < 4; ++r) c[r] = fmaf(hipemu_from<float>(t[(4 * ablk + r) * 16 + 0]), bv, c[r]);
    return c;
}
#define __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, x, y, z) hipemu_mfma4((a), (b), (c), (x), (y))


// ---- bit casts, DPP / permlane / bpermute (semantics: cdna_hip_programming.md T12/T21, ISA ch. DPP) ----------------------
static inline unsigned __float_as_uint(float f) { return hipemu_bits(f); }
static inline int __float_as_int(float f) { return (int)hipemu_bits(f); }
static inline float __uint_as_float(unsigned u) { return hipemu_from<float>(u); }
static inline float __int_as_float(int i) { return hipemu_from<float>((uint32_t)i); }
typedef unsigned hipemu_u2 __attribute__((ext_vector_type(2)));
// v_permlane32_swap vdst, src: lanes 32..63 of vdst <-> lanes 0..31 of src; returns {new vdst, new src}
static inline hipemu_u2 __builtin_amdgcn_permlane32_swap(unsigned a, unsigned b, bool, bool) {
    uint32_t w[2] = {a, b};
    const uint32_t* t = hipemu_wave_exchange(w, 2);
    const int l = hipemu_lane();
    hipemu_u2 r;
    r[0] = l < 32 ? a : t[(l - 32) * 16 + 1];
    r[1] = l < 32 ? t[(l + 32) * 16 + 0] : b;
    return r;
}
// v_permlane16_swap vdst, src: odd rows (of 16 lanes) of vdst <-> even rows of src
static inline hipemu_u2 __builtin_amdgcn_permlane16_swap(unsigned a, unsigned b, bool, bool) {
    uint32_t w[2] = {a, b};
    const uint32_t* t = hipemu_wave_exchange(w, 2);
    const int l = hipemu_lane();
    const bool odd = (l >> 4) & 1;
    hipemu_u2 r;
    r[0] = odd ? t[(l - 16) * 16 + 1] : a;
    r[1] = odd ? b : t[(l + 16) * 16 + 0];
    return r;
}
// v_mov_b32_dpp: quad_perm, row_shl/shr/ror, row_mirror, row_half_mirror; row_mask bit (lane >> 4) and bank_mask bit
// ((lane >> 2) & 3: a bank is four CONSECUTIVE lanes of a row) enable the write, disabled lanes keep `old`
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    uint32_t w = (uint32_t)src;
    const uint32_t* t = hipemu_wave_exchange(&w, 1);
    const int l = hipemu_lane(), row = l & ~15, p = l & 15;
    if (!((row_mask >> ((l >> 4) & 3)) & 1) || !((bank_mask >> ((l >> 2) & 3)) & 1)) return old;
    if (ctrl == 0x130 || ctrl == 0x138) {   // wave_shl:1 / wave_shr:1 (gfx9): whole-wave shifts by one lane
        const int src_lane = ctrl == 0x130 ? l + 1 : l - 1;
        if (src_lane < 0 || src_lane > 63) return bound_ctrl ? 0 : old;
        return (int)t[src_lane * 16];
    }
    int sp = -1;   // source position inside the row; -1 = out of range
    if (ctrl >= 0 && ctrl <= 0xff) sp = (p & ~3) | ((ctrl >> (2 * (p & 3))) & 3);
    else if (ctrl >= 0x101 && ctrl <= 0x10f) { sp = p + (ctrl - 0x100); if (sp > 15) sp = -1; }
    else if (ctrl >= 0x111 && ctrl <= 0x11f) { sp = p - (ctrl - 0x110); }
    else if (ctrl >= 0x121 && ctrl <= 0x12f) sp = (p - (ctrl - 0x120)) & 15;
    else if (ctrl == 0x140) sp = 15 - p;
    else if (ctrl == 0x141) sp = (p & 8) | (7 - (p & 7));
    else { fprintf(stderr, "hipemu: unsupported dpp_ctrl 0x%x\n", ctrl); abort(); }
    if (sp < 0) return bound_ctrl ? 0 : old;
    return (int)t[(row + sp) * 16];
}
static inline int __builtin_amdgcn_ds_bpermute(int addr, int data) {
    uint32_t w = (uint32_t)data;
    const uint32_t* t = hipemu_wave_exchange(&w, 1);
    return (int)t[((addr >> 2) & 63) * 16];
}

static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline double atomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }

static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __frcp_rn(float a) { return 1.0f / a; }
static inline float __builtin_amdgcn_rcpf(float a) { return 1.0f / a; }
static inline float __builtin_amdgcn_exp2f(float a) { return exp2f(a); }
static inline float __builtin_amdgcn_logf(float a) { return log2f(a); }
using std::max;
using std::min;

// wave-scope sync used by howl_common.hip.h::wave_lds_sync()
#define __builtin_amdgcn_fence(order, scope) ((void)0)
static inline void __builtin_amdgcn_wave_barrier() { hipemu_wave_exchange(nullptr, 0); }
