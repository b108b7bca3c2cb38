// Diagnostic (not part of the library): what does "the last block to finish folds everybody's partials" cost on gfx950?
//   mode 0: no arrival protocol (lower bound)                mode 1: every thread __threadfence(), thread 0 atomicAdd
//   mode 2: __syncthreads(), thread 0 __threadfence() + add   mode 3: partials as agent-scope relaxed atomic stores
//                                                                     (write-through), __syncthreads(), thread 0 relaxed add,
//                                                                     the last block reads them with agent-scope atomic loads
// Each block first writes `dirty` floats per thread of ordinary output (what a convolution does), then one row of 64 partials.
// The last block checks the fold against the closed form; mismatches are counted (a stale read shows up there).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fence_probe tools/fence_probe.hip && /tmp/fence_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE, int FOLD>
__global__ __launch_bounds__(256) void probe(float* out, int dirty, float* part, unsigned* counter, float* result, unsigned* bad,
                                             int iter) {
    __shared__ unsigned ticket;
    const int tid = threadIdx.x;
    const size_t base = ((size_t)blockIdx.x * 256 + tid) * dirty;
    for (int i = 0; i < dirty; ++i) out[base + i] = (float)(iter + i);
    const float v = (float)((blockIdx.x + iter) & 1023);
    if (tid < 64) {
        if (MODE == 3)
            __hip_atomic_store(&part[(size_t)blockIdx.x * 64 + tid], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            part[(size_t)blockIdx.x * 64 + tid] = v;
    }
    if (MODE == 0) return;
    if (MODE == 1) __threadfence();
    __syncthreads();
    if (tid == 0) {
        if (MODE == 2) __threadfence();
        const unsigned t = MODE == 3 ? __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                     : atomicAdd(counter, 1u);
        if (t == gridDim.x - 1) *counter = 0u;
        ticket = t;
    }
    __syncthreads();
    if (ticket != gridDim.x - 1) return;
    if (MODE != 3) __threadfence();
    if (FOLD == 0) {
        if (tid == 0) result[0] = 1.0f;
        return;
    }
    // fold: lane = column, the 4 waves take rows wave, wave+4, ..., eight loads in flight (what the library does)
    __shared__ double red[4][64];
    const int lane = tid & 63, rg = tid >> 6;
    const int R = gridDim.x;
    double a = 0.0;
    for (int k0 = rg; k0 < R; k0 += 32) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 4 * u < R ? k0 + 4 * u : R - 1;
            v[u] = MODE == 3 ? __hip_atomic_load(&part[(size_t)k * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                             : part[(size_t)k * 64 + lane];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) a += k0 + 4 * u < R ? (double)v[u] : 0.0;
    }
    red[rg][lane] = a;
    __syncthreads();
    if (rg == 0) {
        const double s = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
        double want = 0.0;
        for (int k = 0; k < R; ++k) want += (double)((k + iter) & 1023);
        result[lane] = (float)s;
        if (s != want) atomicAdd(bad, 1u);
    }
}

template <int MODE, int FOLD>
void run(int blocks, int dirty, float* out, float* part, unsigned* counter, float* result, unsigned* bad) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipMemset(bad, 0, 4);
    hipMemset(counter, 0, 4);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((probe<MODE, FOLD>), dim3(blocks), dim3(256), 0, 0, out, dirty, part, counter, result, bad, i);
    hipEventRecord(e0, 0);
    const int iters = 200;
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL((probe<MODE, FOLD>), dim3(blocks), dim3(256), 0, 0, out, dirty, part, counter, result, bad, i + 5);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned hbad = 0;
    hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
    printf("mode %d fold %d blocks %5d dirty %3d floats/thread: %8.2f us per launch   mismatching folds %u\n", MODE, FOLD, blocks, dirty,
           ms * 1000.0f / iters, hbad);
}

int main() {
    float *out, *part, *result;
    unsigned *counter, *bad;
    hipMalloc(&out, (size_t)8192 * 256 * 64 * 4);
    hipMalloc(&part, (size_t)8192 * 64 * 4);
    hipMalloc(&result, 64 * 4);
    hipMalloc(&counter, 4);
    hipMalloc(&bad, 4);
    for (int dirty : {4, 32})
        for (int blocks : {256, 1024, 4096}) {
            run<0, 0>(blocks, dirty, out, part, counter, result, bad);
            run<1, 0>(blocks, dirty, out, part, counter, result, bad);
            run<2, 0>(blocks, dirty, out, part, counter, result, bad);
            run<3, 0>(blocks, dirty, out, part, counter, result, bad);
            run<1, 1>(blocks, dirty, out, part, counter, result, bad);
            run<2, 1>(blocks, dirty, out, part, counter, result, bad);
            run<3, 1>(blocks, dirty, out, part, counter, result, bad);
        }
    return 0;
}
