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
constexpr int CTC_CHUNK = 128;      // time steps per LDS window (4 x 128 x 65 floats = 133 KB)
constexpr int CTC_MAX_T = 8192;     // 82 s of 10-ms frames; nothing in the kernel depends on it but 32-bit row offsets

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

// What one wave keeps of the utterance while it walks the time axis in windows of <= CTC_CHUNK rows.
struct CtcLane {
    int lane, lab, S, C;
    bool live, skip_a, skip_b, want_grad;
    float *lpbuf, *abuf, *bbuf, *qbuf;     // [rows][RP] each: log-softmax, alpha, beta, class posteriors of the window
    const int* labbuf;
};

// phase 1 for the rows [t0, t0 + len) of an utterance: logits -> LDS (flat, independent loads), then lane r turns the
// window's rows r, r + 64 into log-softmax rows in place (no cross-lane reduction) and clears their posterior rows.
// A row's bits do not depend on the window it is staged in (a row is recomputed when the backward sweep returns to it).
__device__ __forceinline__ void ctc_stage_rows(const CtcLane& w, const float* __restrict__ zb, long st_t, int t0, int len) {
    const int C = w.C, n = len * C;
    for (int i0 = w.lane; i0 < n; i0 += 4 * 64) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 64 * u < n ? i0 + 64 * u : n - 1;
            const int r = i / C;
            v[u] = zb[(size_t)(t0 + r) * st_t + (i - r * C)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 64 * u;
            if (i < n) {
                const int r = i / C;
                w.lpbuf[r * RP + (i - r * C)] = v[u];
            }
        }
    }
    __syncthreads();
    for (int r = w.lane; r < len; r += 64) {
        float* row = w.lpbuf + r * RP;
        float m = row[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, row[c]);
        float se = 0.0f;
        for (int c = 0; c < C; ++c) se += expf(row[c] - m);
        const float lse = m + logf(se);
        for (int c = 0; c < C; ++c) row[c] -= lse;
        if (w.want_grad)
            for (int c = 0; c < C; ++c) w.qbuf[r * RP + c] = 0.0f;
    }
    __syncthreads();
}

// one step of either recursion: the neighbour states by DPP wave shifts (wave_shr:1 / wave_shl:1, gfx9): a register move
// each, where the ds_bpermute behind __shfl was a round trip through the LDS pipe (~100 cycles) on the serial path of every
// time step; lanes the shift leaves without a source (0 / 63, and 1 / 62 on the second hop) are masked
__device__ __forceinline__ float ctc_alpha_step(const CtcLane& w, float a, float lpa) {
    const float a1 = wave_shr1(a), a2 = wave_shr1(a1);
    const float va = lse3(a, w.lane >= 1 ? a1 : -INFINITY, w.skip_a ? a2 : -INFINITY) + lpa;
    return w.live ? va : -INFINITY;
}
__device__ __forceinline__ float ctc_beta_step(const CtcLane& w, float bt, float lpb) {
    const float b1 = wave_shl1(bt), b2 = wave_shl1(b1);
    const float vb = lse3(bt, w.lane + 1 < w.S ? b1 : -INFINITY, w.skip_b ? b2 : -INFINITY) + lpb;
    return w.live ? vb : -INFINITY;
}

// phase 3 for a window whose lp / alpha / beta rows are in LDS: (a) lane = time step, walk the states of the row:
// gamma_t(s) = exp(alpha + beta - lp[l'_s] + nll) goes to its class; the even states are all the blank (kept in a register),
// the odd ones add to their label's slot; (b) flat over (row, class), coalesced stores
__device__ __forceinline__ void ctc_window_grad(const CtcLane& w, int t0, int len, int blank, float nll, float scale,
                                                float* __restrict__ db, long dst_t) {
    __syncthreads();
    const int S = w.S, C = w.C;
    for (int r = w.lane; r < len; r += 64) {
        const float* ar = w.abuf + r * RP;
        const float* br = w.bbuf + r * RP;
        const float* lr = w.lpbuf + r * RP;
        float* qr = w.qbuf + r * RP;
        const float lpblank = lr[blank];
        float qblank = 0.0f;
        for (int s2 = 0; s2 < S; s2 += 2) qblank += expf(ar[s2] + br[s2] - lpblank + nll);
        for (int s2 = 1; s2 < S; s2 += 2) {
            const int c = w.labbuf[s2];
            qr[c] += expf(ar[s2] + br[s2] - lr[c] + nll);
        }
        qr[blank] += qblank;
    }
    __syncthreads();
    for (int i = w.lane; i < len * C; i += 64) {
        const int r = i / C, c = i - r * C;
        db[(size_t)(t0 + r) * dst_t + c] = (expf(w.lpbuf[r * RP + c]) - w.qbuf[r * RP + c]) * scale;
    }
}

// The time axis in windows of `tc` rows (tc = min(T, CTC_CHUNK): one window = the whole utterance up to 128 frames, the
// round-1..5 kernel).  Longer utterances (whole clips: AudioSequenceBatchifier, batchifier.py:14-34): windows 0 .. n-2
// run the alpha recursion alone and leave their rows in the caller's workspace ([t][64] per utterance: the lane that
// wrote a word is the lane that reads it back); the LAST window runs alpha (from the carried state) and beta interleaved
// as before and takes its gradient rows; then windows n-2 .. 0 are staged again (log-softmax recomputed, alpha from the
// workspace) for the beta recursion (carried in a register) and their gradient rows.  2 T_b - (last window) dependent steps.
__global__ __launch_bounds__(64) void ctc_kernel(const float* __restrict__ logits, long st_t, long st_b, int T, int B, int C,
                                                 const long long* __restrict__ targets, long tgt_stride,
                                                 const long long* __restrict__ input_lengths,
                                                 const long long* __restrict__ target_lengths, int blank,
                                                 float* __restrict__ nll_out, float* __restrict__ dlogits, long dst_t,
                                                 long dst_b, int tc, float* __restrict__ alpha_ws) {
    HIP_DYNAMIC_SHARED(float, lds)
    int* labbuf = reinterpret_cast<int*>(lds + 4 * RP * tc);   // [64]
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
    CtcLane w;
    w.lane = lane, w.lab = lab, w.S = S, w.C = C;
    w.live = live, w.want_grad = want_grad;
    w.skip_a = live && (lane & 1) && lane >= 2 && lab != lab_m2;
    w.skip_b = (lane & 1) && lane + 2 < S && lab != lab_p2;
    w.lpbuf = lds, w.abuf = lds + RP * tc, w.bbuf = lds + 2 * RP * tc, w.qbuf = lds + 3 * RP * tc;
    w.labbuf = labbuf;
    const float* zb = logits + (size_t)b * st_b;
    float* aws = alpha_ws ? alpha_ws + (size_t)b * T * 64 : nullptr;
    const int nwin = Tb > tc ? (Tb + tc - 1) / tc : 1;

    // windows 0 .. nwin-2: alpha alone
    float a = -INFINITY, bt = -INFINITY;
    for (int j = 0; j + 1 < nwin; ++j) {
        const int t0 = j * tc;
        ctc_stage_rows(w, zb, st_t, t0, tc);
        for (int k = 0; k < tc; ++k) {
            const float lpa = w.lpbuf[k * RP + lab];
            if (t0 + k == 0) a = (live && lane < 2) ? lpa : -INFINITY;
            else a = ctc_alpha_step(w, a, lpa);
            if (want_grad) aws[(size_t)(t0 + k) * 64 + lane] = a;
        }
        __syncthreads();     // the window's LDS rows are rewritten by the next stage
    }
    // the last window: alpha forward and beta backward, interleaved
    const int tl0 = (nwin - 1) * tc, tlen = Tb - tl0;
    ctc_stage_rows(w, zb, st_t, tl0, tlen);
    for (int k = 0; k < tlen; ++k) {
        const int kb = tlen - 1 - k;
        const float lpa = w.lpbuf[k * RP + lab];
        const float lpb = w.lpbuf[kb * RP + lab];
        if (tl0 + k == 0) a = (live && lane < 2) ? lpa : -INFINITY;
        else a = ctc_alpha_step(w, a, lpa);
        if (k == 0) bt = (live && lane >= S - 2) ? lpb : -INFINITY;
        else if (want_grad) bt = ctc_beta_step(w, bt, lpb);
        w.abuf[k * RP + lane] = a;
        if (want_grad) w.bbuf[kb * RP + lane] = bt;
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
    float* db = dlogits + (size_t)b * dst_b;
    const float scale = 1.0f / ((float)B * (float)(L > 0 ? L : 1));
    ctc_window_grad(w, tl0, tlen, blank, nll, scale, db, dst_t);
    // windows nwin-2 .. 0: beta alone, on the alpha rows the first sweep left
    for (int j = nwin - 2; j >= 0; --j) {
        const int t0 = j * tc;
        __syncthreads();
        for (int k0 = 0; k0 < tc; k0 += 8) {       // independent loads, eight rows in flight
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = aws[(size_t)(t0 + k0 + u) * 64 + lane];
#pragma unroll
            for (int u = 0; u < 8; ++u) w.abuf[(k0 + u) * RP + lane] = v[u];
        }
        ctc_stage_rows(w, zb, st_t, t0, tc);
        for (int k = tc - 1; k >= 0; --k) {
            bt = ctc_beta_step(w, bt, w.lpbuf[k * RP + lab]);
            w.bbuf[k * RP + lane] = bt;
        }
        ctc_window_grad(w, t0, tc, blank, nll, scale, db, dst_t);
    }
    // rows past the utterance's end
    for (int i = lane; i < (T - Tb) * C; i += 64) {
        const int r = i / C, c = i - r * C;
        db[(size_t)(Tb + r) * dst_t + c] = 0.0f;
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
