// Microbenchmark (tools only, not part of the product): what the fp32 MFMA pipe sustains on this box in the
// geometry of conv3x3_mfma_kernel (768-thread workgroups, 108 weight fragments in VGPRs), depending on where the A
// operand comes from and how the LDS reads are scheduled.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_ubench.hip -o /tmp/mfma_ubench && /tmp/mfma_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// SRC 0: A from registers; 1: LDS, lane-linear addresses; 2: LDS, conv address pattern (channel stride cs)
// NACC: accumulator chains per wave; SEP: each chain reads its own A (two M tiles) instead of sharing one
// PIPE: pin "next block's DS reads before this block's MFMAs" with sched_group_barrier
template <int SRC, int NACC, bool SEP, bool PIPE, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float* out, int iters, int cs) {
    __shared__ float lds[24576];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 24576; i += THREADS) lds[i] = 1.0f / (1 + i);
    __syncthreads();
    float w[108];
#pragma unroll
    for (int i = 0; i < 108; ++i) w[i] = 0.001f * (i + lane);
    f32x4 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = {0, 0, 0, 0};
    const int m = lane & 15;
    const float* abase = lds + (SRC == 2 ? (lane >> 4) * cs + (m / 10) * 12 + (m % 10) : lane);
    const int cstep = (SRC == 2) ? 4 * cs : 64;
    int opaque = 0;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(opaque));  // keep the LDS reads inside the loop
        const float* a0 = abase + opaque;
        float cur[NACC][9], nxt[NACC][9];
        if (SRC != 0) {
#pragma unroll
            for (int a = 0; a < (SEP ? NACC : 1); ++a)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) cur[a][tap] = a0[a * 48 + (tap / 3) * 12 + (tap % 3)];
            if (PIPE) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int c0 = 0; c0 < 12; ++c0) {
            a0 += cstep;
            if (SRC != 0 && c0 < 11) {
#pragma unroll
                for (int a = 0; a < (SEP ? NACC : 1); ++a)
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) nxt[a][tap] = a0[a * 48 + (tap / 3) * 12 + (tap % 3)];
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int a = 0; a < NACC; ++a) {
                    const float av = (SRC == 0) ? w[(c0 * 9 + tap + 7 + a) % 108] : cur[SEP ? a : 0][tap];
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, w[c0 * 9 + tap], acc[a], 0, 0, 0);
                }
            if (SRC != 0) {
#pragma unroll
                for (int a = 0; a < (SEP ? NACC : 1); ++a)
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) cur[a][tap] = nxt[a][tap];
                if (PIPE) {
                    __builtin_amdgcn_sched_group_barrier(0x100, SEP ? 5 * NACC : 5, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 9 * NACC, 0);
                }
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    out[blockIdx.x * THREADS + tid] = s;
}

template <int SRC, int NACC, bool SEP, bool PIPE, int THREADS>
void run(const char* name, int cs) {
    float* out;
    hipMalloc(&out, 512 * 1024 * 4);
    const int iters = 200;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256;
    hipLaunchKernelGGL((k<SRC, NACC, SEP, PIPE, THREADS>), dim3(grid), dim3(THREADS), 0, 0, out, 10, cs);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SRC, NACC, SEP, PIPE, THREADS>), dim3(grid), dim3(THREADS), 0, 0, out, iters, cs);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = 2048.0 * 108 * NACC * iters * (THREADS / 64.0) * grid;
    printf("%-58s %7.3f ms %7.1f TFLOP/s\n", name, ms, flops / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main() {
    run<0, 1, false, false, 768>("12 waves: regs, 1 chain", 0);
    run<0, 2, false, false, 768>("12 waves: regs, 2 chains", 0);
    run<1, 1, false, false, 768>("12 waves: lds linear, 1 chain, jit reads", 0);
    run<1, 1, false, true, 768>("12 waves: lds linear, 1 chain, pipelined reads", 0);
    run<2, 1, false, false, 768>("12 waves: lds conv cs=337, 1 chain, jit reads", 337);
    run<2, 1, false, true, 768>("12 waves: lds conv cs=337, 1 chain, pipelined", 337);
    run<2, 2, false, true, 768>("12 waves: lds conv, 2 chains sharing A, pipelined", 337);
    run<2, 2, true, false, 768>("12 waves: lds conv, 2 chains own A, jit", 337);
    run<2, 2, true, true, 768>("12 waves: lds conv, 2 chains own A, pipelined", 337);
    run<2, 1, false, true, 512>("8 waves: lds conv, 1 chain, pipelined", 337);
    run<2, 2, true, true, 512>("8 waves: lds conv, 2 chains own A, pipelined", 337);
    run<2, 2, true, true, 256>("4 waves: lds conv, 2 chains own A, pipelined", 337);
    run<2, 4, true, true, 256>("4 waves: lds conv, 4 chains own A, pipelined", 337);
    return 0;
}
