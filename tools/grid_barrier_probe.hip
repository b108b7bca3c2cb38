// Timing skeleton for "one cooperative launch for conv1..conv6, layers separated by a grid-wide arrival barrier" (VERDICT r5
// item 1b): what a layer boundary costs INSIDE a kernel of the 3x3 convolutions' shape (256 workgroups x 768 threads, 145 KB of
// dynamic LDS, one workgroup per CU) against what it costs as a kernel boundary.  No convolution arithmetic: per layer a
// workgroup (1) requests the next layer's 78 KB of packed weights, (2) publishes its row of 96 BatchNorm partial sums with
// device-scope write-through stores, (3) arrives (vmcnt(0), block barrier, one relaxed device-scope fetch_add), (4) the LAST
// arriver folds the 256 rows in fp64 and publishes 96 statistics + a flag while everybody else polls the flag, (5) every
// workgroup reads the statistics, stores the weights to LDS and goes on.  Variants: fold by everybody after the barrier
// (what the kernels' prologues do today); no weights; the same work as L separate launches (prologue-only kernels).
//     hipcc --offload-arch=gfx950 -O3 -o build/grid_barrier_probe tools/grid_barrier_probe.hip && build/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));              \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

constexpr int THREADS = 768, CP = 48, WFLOATS = 3 * 102 * 64;   // 78,336 B of packed weights per layer
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct Args {
    const float* weights;   // [L][WFLOATS]
    float* part;            // [2][nwg][2 * CP]
    float* stats;           // [L][2 * CP]
    unsigned* counter;      // [L]
    unsigned* flag;         // [1]
    float* sink;
    int layers, nwg, mode;  // mode bit 0: last arriver folds (else everybody), bit 1: stage weights, bit 2: XCD-local pre-barrier
    unsigned epoch;         // flag value base of this launch (flags are never reset)
};

// fp64 column sums of the nwg partial rows, 96 columns: thread (c = tid % 96, g = tid / 96) sums rows g, g + 8, ...; LDS meeting
__device__ __forceinline__ void fold_rows(const float* part, int nwg, float* lds_red, float* out96, bool publish, float* stats) {
    const int tid = threadIdx.x, c = tid % 96, g = tid / 96;
    double acc = 0.0;
    for (int r0 = g; r0 < nwg; r0 += 8 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + 8 * u < nwg ? r0 + 8 * u : nwg - 1;
            v[u] = ld_agent(part + (size_t)r * 96 + c);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += r0 + 8 * u < nwg ? (double)v[u] : 0.0;
    }
    reinterpret_cast<double*>(lds_red)[g * 96 + c] = acc;
    __syncthreads();
    if (tid < 96) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += reinterpret_cast<double*>(lds_red)[k * 96 + tid];
        const float s = (float)(t / (double)(nwg * 270));
        out96[tid] = s;
        if (publish) st_agent(stats + tid, s);
    }
}

// bounded poll (a probe must never hang the box: ~2 s, then it gives up and flags the run)
__device__ __forceinline__ void spin_until(unsigned* flag, unsigned want, float* sink) {
    for (long n = 0; n < (1L << 24); ++n) {
        if ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) >= 0) return;
        __builtin_amdgcn_s_sleep(2);
    }
    sink[1] = 1.0f;
}

__global__ __launch_bounds__(THREADS) void chain_kernel(Args a) {
    extern __shared__ float lds[];
    float* wl = lds;                         // [WFLOATS]
    float* red = lds + WFLOATS;              // 8 * 96 doubles
    float* st = red + 2 * 8 * 96;            // [96]
    __shared__ unsigned ticket;
    const int tid = threadIdx.x, wg = blockIdx.x;
    float carry = 0.0f;
    for (int l = 0; l < a.layers; ++l) {
        float4 wv[7];
        if (a.mode & 2) {
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int i = tid + j * THREADS;
                wv[j] = i < WFLOATS / 4 ? reinterpret_cast<const float4*>(a.weights + (size_t)l * WFLOATS)[i] : make_float4(0, 0, 0, 0);
            }
        }
        float* part = a.part + (size_t)(l & 1) * a.nwg * 96;
        if (tid < 96) st_agent(part + (size_t)wg * 96 + tid, 1.0f + carry * 1e-30f + (float)tid);
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
        __syncthreads();
        if (tid == 0) ticket = __hip_atomic_fetch_add(a.counter + l, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const bool last = ticket == (unsigned)a.nwg - 1;
        const unsigned want = a.epoch + (unsigned)l + 1u;
        if (a.mode & 1) {
            if (last) {
                fold_rows(part, a.nwg, red, st, true, a.stats + (size_t)l * 96);
                __builtin_amdgcn_s_waitcnt(0x0F70);
                __syncthreads();
                if (tid == 0) {
                    __hip_atomic_store(a.counter + l, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(a.flag, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                if (tid == 0) spin_until(a.flag, want, a.sink);
                __syncthreads();
                if (tid < 96) st[tid] = ld_agent(a.stats + (size_t)l * 96 + tid);
            }
        } else {
            if (last && tid == 0) {
                __hip_atomic_store(a.counter + l, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.flag, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tid == 0) spin_until(a.flag, want, a.sink);
            __syncthreads();
            fold_rows(part, a.nwg, red, st, false, nullptr);
        }
        if (a.mode & 2) {
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int i = tid + j * THREADS;
                if (i < WFLOATS / 4) reinterpret_cast<float4*>(wl)[i] = wv[j];
            }
        }
        __syncthreads();
        carry += st[tid % 96] + wl[(tid * 17) % WFLOATS];
    }
    if (carry == 12345.678f) a.sink[0] = carry;
}

// the same per-layer work as its own launch: weights + everybody folds the previous launch's partial rows + writes its own row
__global__ __launch_bounds__(THREADS) void layer_kernel(Args a, int l) {
    extern __shared__ float lds[];
    float* wl = lds;
    float* red = lds + WFLOATS;
    float* st = red + 2 * 8 * 96;
    const int tid = threadIdx.x, wg = blockIdx.x;
    if (a.mode & 8) return;      // empty launch
    float4 wv[7];
    if (a.mode & 2) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int i = tid + j * THREADS;
            wv[j] = i < WFLOATS / 4 ? reinterpret_cast<const float4*>(a.weights + (size_t)l * WFLOATS)[i] : make_float4(0, 0, 0, 0);
        }
    }
    fold_rows(a.part + (size_t)((l + 1) & 1) * a.nwg * 96, a.nwg, red, st, false, nullptr);
    if (a.mode & 2) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int i = tid + j * THREADS;
            if (i < WFLOATS / 4) reinterpret_cast<float4*>(wl)[i] = wv[j];
        }
    }
    __syncthreads();
    const float carry = st[tid % 96] + wl[(tid * 17) % WFLOATS];
    if (tid < 96) a.part[(size_t)(l & 1) * a.nwg * 96 + (size_t)wg * 96 + tid] = 1.0f + carry * 1e-30f + (float)tid;
}

int main(int argc, char** argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 256, L = 6, REP = 200;
    Args a{};
    float *w, *part, *stats, *sink;
    unsigned *counter, *flag;
    CK(hipMalloc(&w, (size_t)L * WFLOATS * 4));
    CK(hipMemset(w, 0, (size_t)L * WFLOATS * 4));
    CK(hipMalloc(&part, (size_t)2 * nwg * 96 * 4));
    CK(hipMemset(part, 0, (size_t)2 * nwg * 96 * 4));
    CK(hipMalloc(&stats, (size_t)L * 96 * 4));
    CK(hipMalloc(&sink, 8));
    CK(hipMemset(sink, 0, 8));
    CK(hipMalloc(&counter, L * 4));
    CK(hipMemset(counter, 0, L * 4));
    CK(hipMalloc(&flag, 4));
    CK(hipMemset(flag, 0, 4));
    a.weights = w, a.part = part, a.stats = stats, a.counter = counter, a.flag = flag, a.sink = sink, a.layers = L, a.nwg = nwg;
    const size_t lds = (size_t)(WFLOATS + 2 * 8 * 96 + 96) * 4 + 66000;      // ~145 KB like the convolutions: one workgroup per CU
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    unsigned epoch = 0;
    auto time_chain = [&](int mode, const char* what) {
        a.mode = mode;
        for (int pass = 0; pass < 2; ++pass) {
            CK(hipEventRecord(e0));
            for (int r = 0; r < REP; ++r) {
                a.epoch = epoch;
                epoch += L;
                hipLaunchKernelGGL(chain_kernel, dim3(nwg), dim3(THREADS), lds, 0, a);
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (pass) printf("%-64s %7.2f us per launch of %d layers = %6.2f us per layer\n", what, ms * 1e3 / REP, L, ms * 1e3 / REP / L);
        }
    };
    auto time_layers = [&](int mode, const char* what) {
        a.mode = mode;
        for (int pass = 0; pass < 2; ++pass) {
            CK(hipEventRecord(e0));
            for (int r = 0; r < REP; ++r)
                for (int l = 0; l < L; ++l) hipLaunchKernelGGL(layer_kernel, dim3(nwg), dim3(THREADS), lds, 0, a, l);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (pass) printf("%-64s %7.2f us per %d launches        = %6.2f us per layer\n", what, ms * 1e3 / REP, L, ms * 1e3 / REP / L);
        }
    };
    printf("%d workgroups x %d threads, %zu B of LDS\n", nwg, THREADS, lds);
    time_layers(8, "separate launches, empty kernels");
    time_layers(0, "separate launches: everybody folds 256 rows");
    time_layers(2, "separate launches: fold + 78 KB of weights (today's prologue)");
    time_chain(0, "one launch: barrier, everybody folds");
    time_chain(1, "one launch: last arriver folds, others poll + read 96");
    time_chain(2, "one launch: barrier, everybody folds, weights under the wait");
    time_chain(3, "one launch: last arriver folds, weights under the wait");
    CK(hipDeviceSynchronize());
    float hs[2];
    CK(hipMemcpy(hs, sink, 8, hipMemcpyDeviceToHost));
    printf(hs[1] != 0.0f ? "A POLL GAVE UP: the chain timings above are invalid\n" : "done\n");
    return 0;
}
