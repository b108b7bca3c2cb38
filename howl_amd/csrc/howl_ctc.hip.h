// The fused log_softmax + CTC recursion of one utterance on ONE wavefront (moved out of ctc.hip in round 6 so that lstm.hip's fused
// head + CTC kernel can run it on a wave of a larger workgroup); see ctc.hip's header for what it replaces in the reference.
#pragma once
#include <math.h>

#include "howl_common.hip.h"

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

// What one wave keeps of the utterance while it walks the time axis in windows of <= CTC_CHUNK rows.
struct CtcLane {
    int lane, lab, S, C;
    bool live, skip_a, skip_b, want_grad;
    float *lpbuf, *abuf, *bbuf, *qbuf;     // [rows][RP] each: log-softmax, alpha, beta, class posteriors of the window
    const int* labbuf;
};

// (every helper below runs in ONE wavefront and synchronises with wave_lds_sync only -- no workgroup barrier -- so that a wave of a
// larger workgroup can run an utterance's recursion while the other waves do something else: the fused head + CTC kernel of lstm.hip)
// phase 1 for the rows [t0, t0 + len) of an utterance: logits -> LDS (flat, independent loads), then lane r turns the
// window's rows r, r + 64 into log-softmax rows in place (no cross-lane reduction) and clears their posterior rows.
// A row's bits do not depend on the window it is staged in (a row is recomputed when the backward sweep returns to it).
template <int RP>
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
    wave_lds_sync();
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
    wave_lds_sync();
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
template <int RP>
__device__ __forceinline__ void ctc_window_grad(const CtcLane& w, int t0, int len, int blank, float nll, float scale,
                                                float* __restrict__ db, long dst_t) {
    wave_lds_sync();
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
    wave_lds_sync();
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
template <int RP>
__device__ __forceinline__ void ctc_wave(const float* __restrict__ zb, long st_t, int T, int B, int C, const long long* __restrict__ tgt,
                                         int Tb, int L, int blank, float* __restrict__ nll_out, float* __restrict__ db, long dst_t,
                                         int tc, float* __restrict__ aws, float* __restrict__ lds, int lane) {
    int* labbuf = reinterpret_cast<int*>(lds + 4 * RP * tc);   // [64]
    Tb = Tb < 0 ? 0 : (Tb > T ? T : Tb);
    const int S = 2 * L + 1;
    const bool live = lane < S;
    const bool want_grad = db != nullptr;
    // extended labels and the "may skip the blank between two different labels" flags
    int lab = blank;
    if (live && (lane & 1)) lab = (int)tgt[lane >> 1];
    lab &= 63;
    labbuf[lane] = lab;
    // a row pitch below 64 (the fused head kernel: RP = 17 for <= 8 classes and <= 8 labels) holds the live states only
    const bool wr = RP >= 64 || lane < RP;
    const int lab_m2 = __shfl(lab, lane >= 2 ? lane - 2 : lane);
    const int lab_p2 = __shfl(lab, lane + 2 < 64 ? lane + 2 : lane);
    CtcLane w;
    w.lane = lane, w.lab = lab, w.S = S, w.C = C;
    w.live = live, w.want_grad = want_grad;
    w.skip_a = live && (lane & 1) && lane >= 2 && lab != lab_m2;
    w.skip_b = (lane & 1) && lane + 2 < S && lab != lab_p2;
    w.lpbuf = lds, w.abuf = lds + RP * tc, w.bbuf = lds + 2 * RP * tc, w.qbuf = lds + 3 * RP * tc;
    w.labbuf = labbuf;
    const int nwin = Tb > tc ? (Tb + tc - 1) / tc : 1;

    // windows 0 .. nwin-2: alpha alone
    float a = -INFINITY, bt = -INFINITY;
    for (int j = 0; j + 1 < nwin; ++j) {
        const int t0 = j * tc;
        ctc_stage_rows<RP>(w, zb, st_t, t0, tc);
        for (int k = 0; k < tc; ++k) {
            const float lpa = w.lpbuf[k * RP + lab];
            if (t0 + k == 0) a = (live && lane < 2) ? lpa : -INFINITY;
            else a = ctc_alpha_step(w, a, lpa);
            if (want_grad) aws[(size_t)(t0 + k) * 64 + lane] = a;
        }
        wave_lds_sync();     // the window's LDS rows are rewritten by the next stage
    }
    // the last window: alpha forward and beta backward, interleaved
    const int tl0 = (nwin - 1) * tc, tlen = Tb - tl0;
    ctc_stage_rows<RP>(w, zb, st_t, tl0, tlen);
    for (int k = 0; k < tlen; ++k) {
        const int kb = tlen - 1 - k;
        const float lpa = w.lpbuf[k * RP + lab];
        const float lpb = w.lpbuf[kb * RP + lab];
        if (tl0 + k == 0) a = (live && lane < 2) ? lpa : -INFINITY;
        else a = ctc_alpha_step(w, a, lpa);
        if (k == 0) bt = (live && lane >= S - 2) ? lpb : -INFINITY;
        else if (want_grad) bt = ctc_beta_step(w, bt, lpb);
        if (wr) w.abuf[k * RP + lane] = a;
        if (want_grad && wr) w.bbuf[kb * RP + lane] = bt;
    }
    float nll;
    if (Tb > 0) {
        const float l1 = __shfl(a, S - 1), l2 = S > 1 ? __shfl(a, S - 2) : -INFINITY;
        nll = -lse3(l1, l2, -INFINITY);
    } else {
        nll = L == 0 ? 0.0f : INFINITY;
    }
    if (lane == 0) nll_out[0] = nll;
    if (!want_grad) return;
    const float scale = 1.0f / ((float)B * (float)(L > 0 ? L : 1));
    ctc_window_grad<RP>(w, tl0, tlen, blank, nll, scale, db, dst_t);
    // windows nwin-2 .. 0: beta alone, on the alpha rows the first sweep left
    for (int j = nwin - 2; j >= 0; --j) {
        const int t0 = j * tc;
        wave_lds_sync();
        for (int k0 = 0; k0 < tc; k0 += 8) {       // independent loads, eight rows in flight
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = aws[(size_t)(t0 + k0 + u) * 64 + lane];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (wr) w.abuf[(k0 + u) * RP + lane] = v[u];
        }
        ctc_stage_rows<RP>(w, zb, st_t, t0, tc);
        for (int k = tc - 1; k >= 0; --k) {
            bt = ctc_beta_step(w, bt, w.lpbuf[k * RP + lab]);
            if (wr) w.bbuf[k * RP + lane] = bt;
        }
        ctc_window_grad<RP>(w, t0, tc, blank, nll, scale, db, dst_t);
    }
    // rows past the utterance's end
    for (int i = lane; i < (T - Tb) * C; i += 64) {
        const int r = i / C, c = i - r * C;
        db[(size_t)(Tb + r) * dst_t + c] = 0.0f;
    }
}


// ---- two waves per utterance (one window: T <= tc), for a workgroup that has waves to spare (lstm.hip's fused head kernel) ----------
// The alpha and the beta recursion are independent chains; interleaved in one wave they cost ~185 ns per time step, apart on two
// waves ~110 ns each.  Role 0 stages the rows (log-softmax) into `lds`'s lp / q buffers, runs alpha into abuf and takes nll; role 1
// stages ITS OWN copy of the rows (no exchange in front of the recursions) and runs beta into bbuf.  After a WORKGROUP barrier of
// the caller, ctc_pair_grad (role 0's wave) turns alpha, beta and the rows into the gradient.  Same chains, same order of
// operations as ctc_wave: bit-identical nll and gradient.  LDS: [lp | alpha | beta | q | lp of role 1][T][RP] + 64 labels.
template <int RP>
__device__ __forceinline__ CtcLane ctc_pair_lane(int role, int T, int C, const long long* __restrict__ tgt, int L, int blank, bool want_grad,
                                                 float* __restrict__ lds, int lane) {
    int* labbuf = reinterpret_cast<int*>(lds + 5 * (size_t)RP * T);
    const int S = 2 * L + 1;
    const bool live = lane < S;
    int lab = blank;
    if (live && (lane & 1)) lab = (int)tgt[lane >> 1];
    lab &= 63;
    if (role == 0) labbuf[lane] = lab;
    const int lab_m2 = __shfl(lab, lane >= 2 ? lane - 2 : lane);
    const int lab_p2 = __shfl(lab, lane + 2 < 64 ? lane + 2 : lane);
    CtcLane w;
    w.lane = lane, w.lab = lab, w.S = S, w.C = C;
    w.live = live, w.want_grad = want_grad && role == 0;      // (only role 0's staging clears the posterior rows)
    w.skip_a = live && (lane & 1) && lane >= 2 && lab != lab_m2;
    w.skip_b = (lane & 1) && lane + 2 < S && lab != lab_p2;
    w.lpbuf = role == 0 ? lds : lds + 4 * (size_t)RP * T;
    w.abuf = lds + (size_t)RP * T, w.bbuf = lds + 2 * (size_t)RP * T, w.qbuf = lds + 3 * (size_t)RP * T;
    w.labbuf = labbuf;
    return w;
}
// returns nll (role 0; also written to nll_out[0]) -- role 1 returns 0
template <int RP>
__device__ __forceinline__ float ctc_pair_recursion(int role, const float* __restrict__ zb, long st_t, int T, int C,
                                                    const long long* __restrict__ tgt, int Tb, int L, int blank,
                                                    float* __restrict__ nll_out, float* __restrict__ lds, int lane) {
    Tb = Tb < 0 ? 0 : (Tb > T ? T : Tb);
    const CtcLane w = ctc_pair_lane<RP>(role, T, C, tgt, L, blank, true, lds, lane);
    const bool wr = RP >= 64 || lane < RP;
    ctc_stage_rows<RP>(w, zb, st_t, 0, Tb);
    if (role == 0) {
        float a = -INFINITY;
        for (int k = 0; k < Tb; ++k) {
            const float lpa = w.lpbuf[k * RP + w.lab];
            if (k == 0) a = (w.live && lane < 2) ? lpa : -INFINITY;
            else a = ctc_alpha_step(w, a, lpa);
            if (wr) w.abuf[k * RP + lane] = a;
        }
        float nll;
        if (Tb > 0) {
            const float l1 = __shfl(a, w.S - 1), l2 = w.S > 1 ? __shfl(a, w.S - 2) : -INFINITY;
            nll = -lse3(l1, l2, -INFINITY);
        } else {
            nll = L == 0 ? 0.0f : INFINITY;
        }
        if (lane == 0) nll_out[0] = nll;
        return nll;
    }
    float bt = -INFINITY;
    for (int k = 0; k < Tb; ++k) {
        const int kb = Tb - 1 - k;
        const float lpb = w.lpbuf[kb * RP + w.lab];
        if (k == 0) bt = (w.live && lane >= w.S - 2) ? lpb : -INFINITY;
        else bt = ctc_beta_step(w, bt, lpb);
        if (wr) w.bbuf[kb * RP + lane] = bt;
    }
    return 0.0f;
}
// role 0's wave, after the workgroup barrier behind both recursions
template <int RP>
__device__ __forceinline__ void ctc_pair_grad(int T, int B, int C, const long long* __restrict__ tgt, int Tb, int L, int blank, float nll,
                                              float* __restrict__ db, long dst_t, float* __restrict__ lds, int lane) {
    Tb = Tb < 0 ? 0 : (Tb > T ? T : Tb);
    CtcLane w = ctc_pair_lane<RP>(0, T, C, tgt, L, blank, true, lds, lane);
    const float scale = 1.0f / ((float)B * (float)(L > 0 ? L : 1));
    ctc_window_grad<RP>(w, 0, Tb, blank, nll, scale, db, dst_t);
    for (int i = lane; i < (T - Tb) * C; i += 64) {
        const int r = i / C, c = i - r * C;
        db[(size_t)(Tb + r) * dst_t + c] = 0.0f;
    }
}

}  // namespace
