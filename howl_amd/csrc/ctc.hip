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
#include "howl_ctc.hip.h"
#include "../../include/howl_hip.h"

namespace {

constexpr int RP = 65;   // LDS row pitch: lanes that walk down a column (lane = time step) hit 64 different banks

// one wavefront per utterance: ctc_wave (howl_ctc.hip.h)
__global__ __launch_bounds__(64) void ctc_kernel(const float* __restrict__ logits, long st_t, long st_b, int T, int B, int C,
                                                 const long long* __restrict__ targets, long tgt_stride,
                                                 const long long* __restrict__ input_lengths,
                                                 const long long* __restrict__ target_lengths, int blank,
                                                 float* __restrict__ nll_out, float* __restrict__ dlogits, long dst_t,
                                                 long dst_b, int tc, float* __restrict__ alpha_ws) {
    HIP_DYNAMIC_SHARED(float, lds)
    const int b = blockIdx.x;
    ctc_wave<RP>(logits + (size_t)b * st_b, st_t, T, B, C, targets + (size_t)b * tgt_stride, (int)input_lengths[b],
                 (int)target_lengths[b], blank, nll_out + b, dlogits ? dlogits + (size_t)b * dst_b : nullptr, dst_t, tc,
                 alpha_ws ? alpha_ws + (size_t)b * T * 64 : nullptr, lds, threadIdx.x);
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

size_t howl_ctc_workspace_floats(int T, int B) {
    return T > CTC_CHUNK && B > 0 ? (size_t)B * (size_t)T * 64 : 0;
}

int howl_ctc_loss(const float* logits, long st_t, long st_b, int T, int B, int C, const long long* targets, long tgt_stride,
                  int max_target_length, const long long* input_lengths, const long long* target_lengths, int blank,
                  float* nll, float* loss, float* dlogits, long dst_t, long dst_b, float* workspace, size_t workspace_floats,
                  hipStream_t stream) {
    HOWL_REQUIRE(logits && targets && input_lengths && target_lengths && nll, "howl_ctc_loss: null pointer");
    HOWL_REQUIRE(B >= 1 && blank >= 0 && blank < C, "howl_ctc_loss: bad shape (B=%d, blank=%d, C=%d)", B, blank, C);
    HOWL_REQUIRE(howl_ctc_supported(T, C, max_target_length),
                 "howl_ctc_loss: T=%d C=%d target length %d outside the kernel's range (T <= %d, C <= %d, targets <= %d)", T,
                 C, max_target_length, CTC_MAX_T, CTC_MAX_C, CTC_MAX_L);
    const bool spills = dlogits != nullptr && T > CTC_CHUNK;     // the loss alone keeps nothing of the alpha rows
    HOWL_REQUIRE(!spills || (workspace && workspace_floats >= howl_ctc_workspace_floats(T, B)),
                 "howl_ctc_loss: T=%d > %d frames with a gradient needs a workspace of howl_ctc_workspace_floats(T, B) = %zu floats "
                 "(got %zu)", T, CTC_CHUNK, howl_ctc_workspace_floats(T, B), workspace ? workspace_floats : (size_t)0);
    const int tc = T < CTC_CHUNK ? T : CTC_CHUNK;
    const size_t lds = ((size_t)4 * tc * RP + 64) * sizeof(float);   // 133 KB from T = 128 on
    static thread_local size_t granted[16] = {};
    if (!howl_raise_lds(reinterpret_cast<const void*>(ctc_kernel), lds, granted, "howl_ctc_loss")) return howl_take_pending_error(), HOWL_E_LAUNCH;
    hipLaunchKernelGGL(ctc_kernel, dim3(B), dim3(64), lds, stream, logits, st_t, st_b, T, B, C, targets, tgt_stride,
                       input_lengths, target_lengths, blank, nll, dlogits, dst_t, dst_b, tc, spills ? workspace : (float*)nullptr);
    if (loss != nullptr)     // NULL: the caller takes the batch mean elsewhere (howl_head_bwd's HowlCtcMean: one launch fewer)
        hipLaunchKernelGGL(ctc_mean_kernel, dim3(1), dim3(256), 0, stream, (const float*)nll, target_lengths, B, loss);
    HOWL_CHECK_LAUNCH("howl_ctc_loss");
    return HOWL_OK;
}

}  // extern "C"
