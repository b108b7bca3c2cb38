// Timing skeleton + accuracy microbenchmark for VERDICT r5 item 5: an fp32 value is exactly the sum of three bf16 values
// (hi + mid + lo, 8 significand bits each, signed residuals), every bf16 x bf16 product is exact in fp32 and
// v_mfma_f32_16x16x32_bf16 accumulates in fp32 -- so the 45 -> 45 convolution's K loop could run on the bf16 matrix pipe
// (16 x the fp32 MFMA rate) in 6 passes (terms hi.hi, hi.mid, mid.hi, hi.lo, mid.mid, lo.hi; dropped: < 2^-24 relative) or 9.
//
// (1) ACCURACY (one wave, real data): a 16 x 16 output tile with K = 432 (9 taps x 48 channels) computed by the fp32 MFMA chain the
//     product uses (v_mfma_f32_16x16x4_f32, 108 steps), by the 6-pass and by the 9-pass split form (small terms in an accumulator
//     of their own), each against the fp64 result on the host.
// (2) TIMING (256 workgroups x 768 threads, LDS-fed, the convolution kernel's wave roles: wave = (cout tile, position group),
//     5 / 4 / 4 / 4 position tiles per group): per "utterance" the K loop of the product (102 k-steps, one ds_read_b32 per
//     operand and MFMA) against the split form (14 k-steps of 32: per step 3 ds_read_b128 of weight fragments + 3 per position
//     tile of activation fragments from a channel-innermost bf16 tile of pitch 112 B, 6 or 9 MFMAs per tile).  No staging, no
//     epilogue: the K loop alone, which is 42 k of the 57 k cycles an utterance costs a workgroup of the forward kernel.
//   hipcc --offload-arch=gfx950 -O3 -o build/bf16x3_ubench tools/bf16x3_ubench.hip && build/bf16x3_ubench
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x)                                                       \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            printf("%s: %s\n", #x, hipGetErrorString(e_));          \
            exit(1);                                                \
        }                                                           \
    } while (0)

__device__ __forceinline__ void split3(float a, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)a;
    const float r1 = a - (float)h;      // exact
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);        // exact residual, rounded once
}

// ---- (1) accuracy: A (16 x K) row-major, B (K x 16) row-major, K = 432; out[3][16][16] -------------------------------------
constexpr int KA = 432;
__global__ __launch_bounds__(64) void accuracy_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    f32x4 acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < KA; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * KA + k0 + g], B[(k0 + g) * 16 + r], acc, 0, 0, 0);
    for (int j = 0; j < 4; ++j) out[0 * 256 + (4 * g + j) * 16 + r] = acc[j];
    for (int passes = 6; passes <= 9; passes += 3) {
        f32x4 big = {0, 0, 0, 0}, small = {0, 0, 0, 0};
        for (int k0 = 0; k0 < KA; k0 += 32) {
            bf16x8 ah, am, al, bh, bm, bl;
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + 8 * g + j;
                const float a = k < KA ? A[r * KA + k] : 0.0f, b = k < KA ? B[k * 16 + r] : 0.0f;
                __bf16 h, m, l;
                split3(a, h, m, l);
                ah[j] = h, am[j] = m, al[j] = l;
                split3(b, h, m, l);
                bh[j] = h, bm[j] = m, bl[j] = l;
            }
            big = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, big, 0, 0, 0);
            small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, small, 0, 0, 0);
            small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, small, 0, 0, 0);
            small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, small, 0, 0, 0);
            small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, small, 0, 0, 0);
            small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, small, 0, 0, 0);
            if (passes == 9) {
                small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bl, small, 0, 0, 0);
                small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bm, small, 0, 0, 0);
                small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bl, small, 0, 0, 0);
            }
        }
        for (int j = 0; j < 4; ++j) out[(passes == 6 ? 1 : 2) * 256 + (4 * g + j) * 16 + r] = big[j] + small[j];
    }
}

// ---- (2) timing -----------------------------------------------------------------------------------------------------------------
constexpr int THREADS = 768, H = 27, WP = 12, CS = 369;     // fp32 tile: [48][(H + 1) * 12 + ...] channel stride = 17 (mod 32)
constexpr int KSTEPS = 102;
constexpr int POS = (H + 2) * WP, PITCH = 56;                // split tile: [plane][position][56 bf16] (48 channels + pad: 112 B)
constexpr int KS32 = 14;                                      // 432 / 32 rounded up

template <int NTW>
__device__ __forceinline__ void f32_loop(const float* tile, const float* wl, int nt, int t0, int lane, int iters, f32x4 (&acc)[NTW]) {
    const float* ap[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        int m = 16 * (t0 + 4 * i) + (lane & 15);
        m = m < H * 10 ? m : H * 10 - 1;
        const int h = m / 10;
        ap[i] = tile + (lane >> 4) * CS + h * WP + (m - h * 10);
    }
    const float* bp0 = wl + nt * KSTEPS * 64 + lane;
    int opaque = 0;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(opaque));
        const float* bp = bp0 + opaque;
        float a[NTW], b;
#pragma unroll
        for (int i = 0; i < NTW; ++i) a[i] = ap[i][opaque];
        b = bp[0];
#pragma unroll 2
        for (int grp = 0; grp < 11; ++grp) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                float na[NTW], nb;
                const int off = tap < 8 ? ((tap + 1) / 3) * WP + ((tap + 1) % 3) + grp * 4 * CS : (grp + 1) * 4 * CS;
#pragma unroll
                for (int i = 0; i < NTW; ++i) na[i] = ap[i][off + opaque];
                nb = bp[(grp * 9 + tap + 1) * 64];
#pragma unroll
                for (int i = 0; i < NTW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NTW; ++i) a[i] = na[i];
                b = nb;
            }
        }
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
            const float bb = bp[(99 + s3) * 64];
#pragma unroll
            for (int i = 0; i < NTW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[i][44 * CS + s3 + opaque], bb, acc[i], 0, 0, 0);
        }
    }
}

template <int NTW, int PASSES>
__device__ __forceinline__ void split_loop(const char* planes, const char* wfr, int nt, int t0, int lane, int iters, f32x4 (&acc)[NTW],
                                           f32x4 (&sm)[NTW]) {
    // lane (row = position m, kg = lane >> 4) reads 8 consecutive channels of one tap: k8 = 4 s + kg -> (tap, channel block of 8)
    int abase[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        int m = 16 * (t0 + 4 * i) + (lane & 15);
        m = m < H * 10 ? m : H * 10 - 1;
        const int h = m / 10;
        abase[i] = (h * WP + (m - h * 10)) * PITCH * 2;
    }
    constexpr int PLANE = POS * PITCH * 2, WPLANE = KS32 * 64 * 16;
    int opaque = 0;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(opaque));
#pragma unroll 2
        for (int s = 0; s < KS32; ++s) {
            const int k8 = 4 * s + (lane >> 4);
            const int tap = k8 / 6, c8 = k8 - 6 * tap;
            const int tapoff = (((tap < 9 ? tap : 8) / 3) * WP + (tap % 3)) * PITCH * 2 + c8 * 16 + opaque;
            bf16x8 bfr[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) bfr[p] = *reinterpret_cast<const bf16x8*>(wfr + p * WPLANE + (s * 64 + lane) * 16 + opaque);
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                bf16x8 afr[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) afr[p] = *reinterpret_cast<const bf16x8*>(planes + p * PLANE + abase[i] + tapoff);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[0], bfr[0], acc[i], 0, 0, 0);
                sm[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[0], bfr[1], sm[i], 0, 0, 0);
                sm[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[1], bfr[0], sm[i], 0, 0, 0);
                sm[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[0], bfr[2], sm[i], 0, 0, 0);
                sm[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[1], bfr[1], sm[i], 0, 0, 0);
                sm[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[2], bfr[0], sm[i], 0, 0, 0);
                if (PASSES == 9) {
                    sm[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[1], bfr[2], sm[i], 0, 0, 0);
                    sm[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[2], bfr[1], sm[i], 0, 0, 0);
                    sm[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[2], bfr[2], sm[i], 0, 0, 0);
                }
            }
        }
    }
}

template <int MODE>      // 0: fp32 chain, 6 / 9: split passes
__global__ __launch_bounds__(THREADS) void timing_kernel(float* __restrict__ out, int iters) {
    extern __shared__ char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = wave % 3, mg = wave / 3;
    for (int i = tid; i < 159000 / 4; i += THREADS) reinterpret_cast<float*>(lds)[i] = 1.0f / (float)(1 + (i & 1023));
    __syncthreads();
    float s = 0.0f;
    if (MODE == 0) {
        const float* wl = reinterpret_cast<const float*>(lds);
        const float* tile = wl + 3 * KSTEPS * 64;
        if (mg == 0) {
            f32x4 acc[5] = {};
            f32_loop<5>(tile, wl, nt, mg, lane, iters, acc);
            for (int i = 0; i < 5; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        } else {
            f32x4 acc[4] = {};
            f32_loop<4>(tile, wl, nt, mg, lane, iters, acc);
            for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        }
    } else {
        const char* wfr = lds;                           // [3 planes][14][64 lanes][16 B]: one cout tile's fragments (see header)
        const char* planes = lds + 3 * KS32 * 64 * 16;   // [3][POS][112 B]
        if (mg == 0) {
            f32x4 acc[5] = {}, sm[5] = {};
            split_loop<5, MODE>(planes, wfr, nt, mg, lane, iters, acc, sm);
            for (int i = 0; i < 5; ++i) s += acc[i][0] + sm[i][1] + acc[i][2] + sm[i][3];
        } else {
            f32x4 acc[4] = {}, sm[4] = {};
            split_loop<4, MODE>(planes, wfr, nt, mg, lane, iters, acc, sm);
            for (int i = 0; i < 4; ++i) s += acc[i][0] + sm[i][1] + acc[i][2] + sm[i][3];
        }
    }
    out[blockIdx.x * THREADS + tid] = s;
}

template <int MODE>
void time_mode(const char* what, float* out) {
    const int iters = 64, REP = 20;
    const size_t lds = 159936;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(timing_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float us1 = 0, usN = 0;
    for (int pass = 0; pass < 2; ++pass) {
        for (int n : {1, iters + 1}) {
            CK(hipEventRecord(e0));
            for (int r = 0; r < REP; ++r) hipLaunchKernelGGL(timing_kernel<MODE>, dim3(256), dim3(THREADS), lds, 0, out, n);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            (n == 1 ? us1 : usN) = ms * 1e3f / REP;
        }
    }
    const double per = (usN - us1) / iters;      // us per utterance-pass of a workgroup, launch and fill costs differenced out
    const double flops = 2.0 * 9 * 45 * 45 * 270;
    printf("%-58s %7.3f us per utterance and workgroup = %6.1f TFLOP/s-equivalent on 256 CUs (fp32 peak 157.3)\n", what, per,
           flops * 256 / per * 1e-6);
}

int main() {
    // ---- accuracy
    std::vector<float> A(16 * KA), B(KA * 16), out(3 * 256);
    srand(7);
    auto rnd = []() { return (float)((rand() / (double)RAND_MAX) * 2.0 - 1.0); };
    for (auto& v : A) v = rnd() * 1.7f;               // activations after BatchNorm: O(1)
    for (auto& v : B) v = rnd() * 0.07f;              // closed-form conv weights: O(0.07)
    float *dA, *dB, *dO;
    CK(hipMalloc(&dA, A.size() * 4));
    CK(hipMalloc(&dB, B.size() * 4));
    CK(hipMalloc(&dO, out.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(accuracy_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dO);
    CK(hipMemcpy(out.data(), dO, out.size() * 4, hipMemcpyDeviceToHost));
    double err[3] = {0, 0, 0}, rms[3] = {0, 0, 0}, ref_max = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double ref = 0;
            for (int k = 0; k < KA; ++k) ref += (double)A[i * KA + k] * (double)B[k * 16 + j];
            ref_max = fmax(ref_max, fabs(ref));
            for (int v = 0; v < 3; ++v) {
                const double e = fabs((double)out[v * 256 + i * 16 + j] - ref);
                err[v] = fmax(err[v], e);
                rms[v] += e * e;
            }
        }
    printf("accuracy, 16 x 16 tile, K = %d, |result| <= %.3f: max |error| vs fp64 (rms)\n", KA, ref_max);
    const char* names[3] = {"fp32 MFMA chain (the product's arithmetic)", "3 x bf16 split, 6 passes", "3 x bf16 split, 9 passes"};
    for (int v = 0; v < 3; ++v) printf("  %-46s %.3e  (%.3e)\n", names[v], err[v], sqrt(rms[v] / 256));
    // ---- timing
    float* sink;
    CK(hipMalloc(&sink, 256 * THREADS * 4));
    time_mode<0>("fp32 chain, 102 k-steps (the product's K loop)", sink);
    time_mode<6>("3 x bf16 split, 6 passes, 14 k-steps of 32", sink);
    time_mode<9>("3 x bf16 split, 9 passes", sink);
    CK(hipDeviceSynchronize());
    return 0;
}
