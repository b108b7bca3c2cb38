// Probe (tools only): operand / result layout and issue rate of v_mfma_f32_4x4x1_16b_f32 on gfx950, the candidate for an LSTM
// recurrence with 4 sequences per workgroup (16 independent 4x4 outer products per instruction).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma4x4_probe.hip -o /tmp/mfma4x4_probe && /tmp/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout(float* out, int which) {
    const int lane = threadIdx.x;
    // which = 0: A = 1 + lane, B = 1 -> D shows the A lane that fed each result; which = 1: the B lane
    f32x4 acc = {0, 0, 0, 0};
    const float a = which == 0 ? 1.0f + lane : 1.0f;
    const float b = which == 1 ? 1.0f + lane : 1.0f;
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = acc[r];
}

// cbsz = 4, abid = 5: every block takes its A operand from block 5 (lanes 20..23)
__global__ void layout_bcast(float* out) {
    const int lane = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f + lane, 1.0f, acc, 4, 5, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = acc[r];
}

template <int NACC>
__global__ __launch_bounds__(1024) void rate(float* out, int iters) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = {0, 0, 0, 0};
    float a = 0.001f * threadIdx.x, b = 1.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* d;
    hipMalloc(&d, 1 << 22);
    float ha[256], hb[256];
    layout<<<1, 64>>>(d, 0);
    hipMemcpy(ha, d, sizeof(ha), hipMemcpyDeviceToHost);
    layout<<<1, 64>>>(d, 1);
    hipMemcpy(hb, d, sizeof(hb), hipMemcpyDeviceToHost);
    printf("lane: D[r = 0..3] = A(lane a) * B(lane b), shown as (a,b)\n");
    for (int lane = 0; lane < 64; ++lane)
        if (lane < 8 || lane % 16 == 0 || lane == 63) {
            printf("lane %2d:", lane);
            for (int r = 0; r < 4; ++r) printf("  (%2d,%2d)", (int)ha[lane * 4 + r] - 1, (int)hb[lane * 4 + r] - 1);
            printf("\n");
        }
    layout_bcast<<<1, 64>>>(d);
    hipMemcpy(ha, d, sizeof(ha), hipMemcpyDeviceToHost);
    bool bc_ok = true;
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) bc_ok = bc_ok && (int)ha[lane * 4 + r] - 1 == 20 + r;
    printf("cbsz=4 abid=5: every D[r] fed by A lane 20 + r: %s (lane 0: %d %d %d %d, lane 63: %d %d %d %d)\n", bc_ok ? "yes" : "NO",
           (int)ha[0] - 1, (int)ha[1] - 1, (int)ha[2] - 1, (int)ha[3] - 1, (int)ha[252] - 1, (int)ha[253] - 1, (int)ha[254] - 1,
           (int)ha[255] - 1);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int dev = 0, cus = 0, khz = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, dev);
    const int iters = 20000;
    auto run = [&](auto kern, int nacc, int threads) {
        kern<<<cus, threads>>>(d, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        kern<<<cus, threads>>>(d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double inst_per_simd = (double)iters * nacc * (threads / 64) / 4.0;
        const double cyc = ms * 1e-3 * khz * 1e3 / inst_per_simd;
        printf("%2d chains x %2d waves/CU: %.3f ms, %.2f cycles per 4x4x1_16b instruction per SIMD (nominal clock %d MHz), %.1f TFLOP/s\n",
               nacc, threads / 64, ms, cyc, khz / 1000, 512.0 * iters * nacc * (threads / 64) * cus / (ms * 1e-3) / 1e12);
    };
    run(rate<1>, 1, 256);
    run(rate<4>, 4, 256);
    run(rate<8>, 8, 256);
    run(rate<4>, 4, 1024);
    run(rate<8>, 8, 1024);
    return 0;
}
