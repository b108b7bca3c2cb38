// Fused log_softmax + CTC loss, forward and backward in one launch (gfx950).
//
// Replaces, in the sequence-model branch of the training loop (training/run/train.py:250-256, 291-296 of the reference),
//     scores = log_softmax(model(...), -1);  loss = CTCLoss(blank)(scores, targets, input_lengths, target_lengths)
// and the autograd backward of the two: the result is the loss and d loss / d logits directly.
//
// One wavefront per utterance: lane s is state s of the extended label sequence l' = (blank, l1, blank, ..., lL, blank),
// S = 2L + 1 <= 63.  The alpha recursion runs forward in time with the two predecessor states fetched by lane shuffles,
// alpha_t and the log-softmax rows stay in LDS, the beta recursion runs backward and emits the gradient row of its time
// step on the way:  d/dz[t][c] = (softmax[t][c] - sum_{s: l'_s = c} gamma_t(s)) / (B * max(L, 1)),
// gamma_t(s) = exp(alpha_t(s) + beta_t(s) - lp[t][l'_s] + nll)  (the state posteriors; rows t >= input_length are zero).
// Same arithmetic as torch's ctc_loss (log-space three-way logsumexp with the running maximum), reduction "mean",
// zero_infinity = False.  Everything is a fixed-order computation: repeated calls are bit-identical.
#include <math.h>

#include "howl_common.hip.h"
#include "../../include/howl_hip.h"

namespace {

constexpr int CTC_MAX_C = 64;
constexpr int CTC_MAX_L = 31;
constexpr int CTC_MAX_T = 128;

__device__ __forceinline__ float wave_max(float v) {
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

// log(exp(a) + exp(b) + exp(c)) with -inf operands allowed
__device__ __forceinline__ float lse3(float a, float b, float c) {
    float m = fmaxf(a, fmaxf(b, c));
    if (m == -INFINITY) m = 0.0f;
    return logf(expf(a - m) + expf(b - m) + expf(c - m)) + m;
}

__global__ __launch_bounds__(64) void ctc_kernel(const float* __restrict__ logits, long st_t, long st_b, int T, int B, int C,
                                                 const long long* __restrict__ targets, long tgt_stride,
                                                 const long long* __restrict__ input_lengths,
                                                 const long long* __restrict__ target_lengths, int blank,
                                                 float* __restrict__ nll_out, float* __restrict__ dlogits, long dst_t,
                                                 long dst_b) {
    HIP_DYNAMIC_SHARED(float, lds)
    float* abuf = lds;            // [T][64] alpha
    float* lpbuf = lds + 64 * T;  // [T][64] log-softmax rows
    const int b = blockIdx.x, lane = threadIdx.x;
    int Tb = (int)input_lengths[b];
    Tb = Tb < 0 ? 0 : (Tb > T ? T : Tb);
    const int L = (int)target_lengths[b];
    const int S = 2 * L + 1;
    const bool live = lane < S;
    // extended labels and the "may skip the blank between two different labels" flags
    int lab = blank;
    if (live && (lane & 1)) lab = (int)targets[(size_t)b * tgt_stride + (lane >> 1)];
    lab &= 63;
    const int lab_m2 = __shfl(lab, lane >= 2 ? lane - 2 : lane);
    const int lab_p2 = __shfl(lab, lane + 2 < 64 ? lane + 2 : lane);
    const bool skip_a = live && (lane & 1) && lane >= 2 && lab != lab_m2;
    const bool skip_b = (lane & 1) && lane + 2 < S && lab != lab_p2;
    const float* zb = logits + (size_t)b * st_b;

    float a = -INFINITY;
    for (int t = 0; t < Tb; ++t) {
        const float z = lane < C ? zb[(size_t)t * st_t + lane] : -INFINITY;
        const float m = wave_max(z);
        const float e = lane < C ? expf(z - m) : 0.0f;
        const float lse = m + logf(wave_sum(e));
        const float lp = lane < C ? z - lse : -INFINITY;
        lpbuf[t * 64 + lane] = lp;
        const float lps = __shfl(lp, lab);
        if (t == 0) {
            a = (live && lane < 2) ? lps : -INFINITY;
        } else {
            const float a1 = __shfl(a, lane >= 1 ? lane - 1 : lane);
            const float a2 = __shfl(a, lane >= 2 ? lane - 2 : lane);
            const float v = lse3(a, lane >= 1 ? a1 : -INFINITY, skip_a ? a2 : -INFINITY) + lps;
            a = live ? v : -INFINITY;
        }
        abuf[t * 64 + lane] = a;
    }
    float nll;
    if (Tb > 0) {
        const float l1 = __shfl(a, S - 1), l2 = S > 1 ? __shfl(a, S - 2) : -INFINITY;
        nll = -lse3(l1, l2, -INFINITY);
    } else {
        nll = L == 0 ? 0.0f : INFINITY;
    }
    if (lane == 0) nll_out[b] = nll;
    if (dlogits == nullptr) return;

    float* db = dlogits + (size_t)b * dst_b;
    const float scale = 1.0f / ((float)B * (float)(L > 0 ? L : 1));
    float bt = -INFINITY;
    for (int t = Tb - 1; t >= 0; --t) {
        const float lp = lpbuf[t * 64 + lane];
        const float lps = __shfl(lp, lab);
        if (t == Tb - 1) {
            bt = (live && lane >= S - 2) ? lps : -INFINITY;
        } else {
            const float b1 = __shfl(bt, lane + 1 < 64 ? lane + 1 : lane);
            const float b2 = __shfl(bt, lane + 2 < 64 ? lane + 2 : lane);
            const float v = lse3(bt, lane + 1 < S ? b1 : -INFINITY, skip_b ? b2 : -INFINITY) + lps;
            bt = live ? v : -INFINITY;
        }
        const float gamma = live ? expf(abuf[t * 64 + lane] + bt - lps + nll) : 0.0f;
        // posterior mass per class: all even states are the blank, the odd ones are looked at one by one
        float q = wave_sum((lane & 1) ? 0.0f : gamma);
        q = lane == (blank & 63) ? q : 0.0f;
        for (int i = 0; i < L; ++i) {
            const float gi = __shfl(gamma, 2 * i + 1);
            const int li = __shfl(lab, 2 * i + 1);
            q += lane == li ? gi : 0.0f;
        }
        if (lane < C) db[(size_t)t * dst_t + lane] = (expf(lp) - q) * scale;
    }
    for (int t = Tb; t < T; ++t)
        if (lane < C) db[(size_t)t * dst_t + lane] = 0.0f;
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
    HOWL_REQUIRE(logits && targets && input_lengths && target_lengths && nll && loss, "howl_ctc_loss: null pointer");
    HOWL_REQUIRE(B >= 1 && blank >= 0 && blank < C, "howl_ctc_loss: bad shape (B=%d, blank=%d, C=%d)", B, blank, C);
    HOWL_REQUIRE(howl_ctc_supported(T, C, max_target_length),
                 "howl_ctc_loss: T=%d C=%d target length %d outside the kernel's range (T <= %d, C <= %d, targets <= %d)", T,
                 C, max_target_length, CTC_MAX_T, CTC_MAX_C, CTC_MAX_L);
    const size_t lds = (size_t)2 * T * 64 * sizeof(float);
    hipLaunchKernelGGL(ctc_kernel, dim3(B), dim3(64), lds, stream, logits, st_t, st_b, T, B, C, targets, tgt_stride,
                       input_lengths, target_lengths, blank, nll, dlogits, dst_t, dst_b);
    hipLaunchKernelGGL(ctc_mean_kernel, dim3(1), dim3(256), 0, stream, (const float*)nll, target_lengths, B, loss);
    HOWL_CHECK_LAUNCH("howl_ctc_loss");
    return HOWL_OK;
}

}  // extern "C"
