// Fused log_softmax + CTC loss, forward and backward in one launch (gfx950).
//
// Replaces, in the sequence-model branch of the training loop (training/run/train.py:250-256, 291-296 of the reference),
//     scores = log_softmax(model(...), -1);  loss = CTCLoss(blank)(scores, targets, input_lengths, target_lengths)
// and the autograd backward of the two: the result is the loss and d loss / d logits directly.
//
// One wavefront per utterance: lane s is state s of the extended label sequence l' = (blank, l1, blank, ..., lL, blank),
// S = 2L + 1 <= 63.  Three phases, all in LDS ([T][65] rows of log-softmax, alpha, beta, class posteriors):
//   1. log_softmax of every row, the rows spread over the lanes (no cross-lane reduction);
//   2. the alpha recursion forward in time and the beta recursion backward in time IN THE SAME LOOP (they are independent,
//      so each hides the other's shuffle / exp / log latency); the two predecessor states come from lane shuffles;
//   3. time steps are independent again and spread over the lanes: gamma_t(s) = exp(alpha_t(s) + beta_t(s) - lp[t][l'_s] + nll),
//      d/dz[t][c] = (softmax[t][c] - sum_{s: l'_s = c} gamma_t(s)) / (B * max(L, 1)); rows t >= input_length are zero.
// Same arithmetic as torch's ctc_loss (log-space three-way logsumexp with the running maximum), reduction "mean",
// zero_infinity = False.  Everything is a fixed-order computation: repeated calls are bit-identical.
#include <math.h>

#include "howl_common.hip.h"
#include "../../include/howl_hip.h"

namespace {

constexpr int CTC_MAX_C = 64;
constexpr int CTC_MAX_L = 31;
constexpr int CTC_MAX_T = 128;

// log(exp(a) + exp(b) + exp(c)) with -inf operands allowed.  This sits on the serial path of the recursions (one wave,
// T dependent steps): the hardware exp2 / log2 instructions (1 ulp) instead of ~150 instructions of library expf / logf.
// The arguments of the exponentials are <= 0 and only the ones near 0 carry weight, so the scaling by log2(e) costs
// nothing measurable (parity with torch's CPU ctc_loss: tests/test_gpu_lstm.py, tests/test_emu_ctc.py).
__device__ __forceinline__ float lse3(float a, float b, float c) {
    constexpr float LOG2E = 1.44269504088896341f, LN2 = 0.693147180559945309f;
    float m = fmaxf(a, fmaxf(b, c));
    if (m == -INFINITY) m = 0.0f;
    const float e = __builtin_amdgcn_exp2f((a - m) * LOG2E) + __builtin_amdgcn_exp2f((b - m) * LOG2E) +
                    __builtin_amdgcn_exp2f((c - m) * LOG2E);
    return __builtin_amdgcn_logf(e) * LN2 + m;
}

__device__ __forceinline__ float wave_shr1(float v) {      // lane i <- lane i - 1 (lane 0 keeps its own)
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_shl1(float v) {      // lane i <- lane i + 1 (lane 63 keeps its own)
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x130, 0xf, 0xf, false));
}

constexpr int RP = 65;   // LDS row pitch: lanes that walk down a column (lane = time step) hit 64 different banks

__global__ __launch_bounds__(64) void ctc_kernel(const float* __restrict__ logits, long st_t, long st_b, int T, int B, int C,
                                                 const long long* __restrict__ targets, long tgt_stride,
                                                 const long long* __restrict__ input_lengths,
                                                 const long long* __restrict__ target_lengths, int blank,
                                                 float* __restrict__ nll_out, float* __restrict__ dlogits, long dst_t,
                                                 long dst_b) {
    HIP_DYNAMIC_SHARED(float, lds)
    float* lpbuf = lds;                  // [T][RP] logits, then log-softmax rows (columns >= C unused)
    float* abuf = lds + RP * T;          // [T][RP] alpha
    float* bbuf = lds + 2 * RP * T;      // [T][RP] beta            (only when the gradient is wanted)
    float* qbuf = lds + 3 * RP * T;      // [T][RP] class posteriors (only when the gradient is wanted)
    int* labbuf = reinterpret_cast<int*>(lds + 4 * RP * T);   // [64]
    const int b = blockIdx.x, lane = threadIdx.x;
    int Tb = (int)input_lengths[b];
    Tb = Tb < 0 ? 0 : (Tb > T ? T : Tb);
    const int L = (int)target_lengths[b];
    const int S = 2 * L + 1;
    const bool live = lane < S;
    const bool want_grad = dlogits != nullptr;
    // extended labels and the "may skip the blank between two different labels" flags
    int lab = blank;
    if (live && (lane & 1)) lab = (int)targets[(size_t)b * tgt_stride + (lane >> 1)];
    lab &= 63;
    labbuf[lane] = lab;
    const int lab_m2 = __shfl(lab, lane >= 2 ? lane - 2 : lane);
    const int lab_p2 = __shfl(lab, lane + 2 < 64 ? lane + 2 : lane);
    const bool skip_a = live && (lane & 1) && lane >= 2 && lab != lab_m2;
    const bool skip_b = (lane & 1) && lane + 2 < S && lab != lab_p2;
    const float* zb = logits + (size_t)b * st_b;

    // phase 1: logits -> LDS (flat, independent loads), then lane t turns rows t, t + 64 into log-softmax rows in place
    const int n = Tb * C;
    for (int i0 = lane; i0 < n; i0 += 4 * 64) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 64 * u < n ? i0 + 64 * u : n - 1;
            const int t = i / C;
            v[u] = zb[(size_t)t * st_t + (i - t * C)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 64 * u;
            if (i < n) {
                const int t = i / C;
                lpbuf[t * RP + (i - t * C)] = v[u];
            }
        }
    }
    __syncthreads();
    for (int t = lane; t < Tb; t += 64) {
        float* row = lpbuf + t * RP;
        float m = row[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, row[c]);
        float se = 0.0f;
        for (int c = 0; c < C; ++c) se += expf(row[c] - m);
        const float lse = m + logf(se);
        for (int c = 0; c < C; ++c) row[c] -= lse;
        if (want_grad)
            for (int c = 0; c < C; ++c) qbuf[t * RP + c] = 0.0f;
    }
    __syncthreads();

    // phase 2: alpha forward and beta backward, interleaved
    float a = -INFINITY, bt = -INFINITY;
    for (int k = 0; k < Tb; ++k) {
        const int tb = Tb - 1 - k;
        const float lpa = lpbuf[k * RP + lab];
        const float lpb = lpbuf[tb * RP + lab];
        if (k == 0) {
            a = (live && lane < 2) ? lpa : -INFINITY;
            bt = (live && lane >= S - 2) ? lpb : -INFINITY;
        } else {
            // the neighbour states by DPP wave shifts (wave_shr:1 / wave_shl:1, gfx9): a register move each, where the ds_bpermute
            // behind __shfl was a round trip through the LDS pipe (~100 cycles) on the serial path of every time step; lanes the
            // shift leaves without a source (0 / 63, and 1 / 62 on the second hop) are masked below as before
            const float a1 = wave_shr1(a), a2 = wave_shr1(a1);
            const float b1 = wave_shl1(bt), b2 = wave_shl1(b1);
            const float va = lse3(a, lane >= 1 ? a1 : -INFINITY, skip_a ? a2 : -INFINITY) + lpa;
            a = live ? va : -INFINITY;
            if (want_grad) {
                const float vb = lse3(bt, lane + 1 < S ? b1 : -INFINITY, skip_b ? b2 : -INFINITY) + lpb;
                bt = live ? vb : -INFINITY;
            }
        }
        abuf[k * RP + lane] = a;
        if (want_grad) bbuf[tb * RP + lane] = bt;
    }
    float nll;
    if (Tb > 0) {
        const float l1 = __shfl(a, S - 1), l2 = S > 1 ? __shfl(a, S - 2) : -INFINITY;
        nll = -lse3(l1, l2, -INFINITY);
    } else {
        nll = L == 0 ? 0.0f : INFINITY;
    }
    if (lane == 0) nll_out[b] = nll;
    if (!want_grad) return;
    __syncthreads();

    // phase 3a: lane = time step.  Walk the states of this row: gamma_t(s) = exp(alpha + beta - lp[l'_s] + nll) goes to
    // its class; the even states are all the blank (kept in a register), the odd ones add to their label's slot.
    for (int t = lane; t < Tb; t += 64) {
        const float* ar = abuf + t * RP;
        const float* br = bbuf + t * RP;
        const float* lr = lpbuf + t * RP;
        float* qr = qbuf + t * RP;
        const float lpblank = lr[blank];
        float qblank = 0.0f;
        for (int s2 = 0; s2 < S; s2 += 2) qblank += expf(ar[s2] + br[s2] - lpblank + nll);
        for (int s2 = 1; s2 < S; s2 += 2) {
            const int c = labbuf[s2];
            qr[c] += expf(ar[s2] + br[s2] - lr[c] + nll);
        }
        qr[blank] += qblank;
    }
    __syncthreads();
    // phase 3b: flat over (t, c), coalesced stores
    float* db = dlogits + (size_t)b * dst_b;
    const float scale = 1.0f / ((float)B * (float)(L > 0 ? L : 1));
    for (int i = lane; i < T * C; i += 64) {
        const int t = i / C, c = i - t * C;
        db[(size_t)t * dst_t + c] = t < Tb ? (expf(lpbuf[t * RP + c]) - qbuf[t * RP + c]) * scale : 0.0f;
    }
}

// loss = mean_b nll_b / max(L_b, 1)  (torch's reduction="mean"), fixed summation order
__global__ __launch_bounds__(256) void ctc_mean_kernel(const float* __restrict__ nll, const long long* __restrict__ target_lengths,
                                                       int B, float* __restrict__ loss) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int b = threadIdx.x; b < B; b += 256) {
        const long long L = target_lengths[b];
        acc += (double)(nll[b] / (float)(L > 0 ? L : 1));
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (float)((((red[0] + red[1]) + red[2]) + red[3]) / (double)B);
}

}  // namespace

extern "C" {

int howl_ctc_supported(int T, int C, int max_target_length) {
    return T >= 1 && T <= CTC_MAX_T && C >= 1 && C <= CTC_MAX_C && max_target_length >= 0 && max_target_length <= CTC_MAX_L;
}

int howl_ctc_loss(const float* logits, long st_t, long st_b, int T, int B, int C, const long long* targets, long tgt_stride,
                  int max_target_length, const long long* input_lengths, const long long* target_lengths, int blank,
                  float* nll, float* loss, float* dlogits, long dst_t, long dst_b, hipStream_t stream) {
    HOWL_REQUIRE(logits && targets && input_lengths && target_lengths && nll, "howl_ctc_loss: null pointer");
    HOWL_REQUIRE(B >= 1 && blank >= 0 && blank < C, "howl_ctc_loss: bad shape (B=%d, blank=%d, C=%d)", B, blank, C);
    HOWL_REQUIRE(howl_ctc_supported(T, C, max_target_length),
                 "howl_ctc_loss: T=%d C=%d target length %d outside the kernel's range (T <= %d, C <= %d, targets <= %d)", T,
                 C, max_target_length, CTC_MAX_T, CTC_MAX_C, CTC_MAX_L);
    const size_t lds = ((size_t)4 * T * RP + 64) * sizeof(float);   // 133 KB at T = 128
    hipFuncSetAttribute(reinterpret_cast<const void*>(ctc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(ctc_kernel, dim3(B), dim3(64), lds, stream, logits, st_t, st_b, T, B, C, targets, tgt_stride,
                       input_lengths, target_lengths, blank, nll, dlogits, dst_t, dst_b);
    if (loss != nullptr)     // NULL: the caller takes the batch mean elsewhere (howl_head_bwd's HowlCtcMean: one launch fewer)
        hipLaunchKernelGGL(ctc_mean_kernel, dim3(1), dim3(256), 0, stream, (const float*)nll, target_lengths, B, loss);
    HOWL_CHECK_LAUNCH("howl_ctc_loss");
    return HOWL_OK;
}

}  // extern "C"
