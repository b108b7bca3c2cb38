// LSTM classifiers for gfx950 (MI355X): single-layer LSTM(40 -> 128) forward + BPTT backward and the Linear heads.
//
// Replaces, for howl/model/rnn.py:41-91 (SequentialLstm "seq-lstm", SimpleLstm "lstm"):
//   pack_padded_sequence + nn.LSTM (rnn.py:65-66,88) .......... howl_lstm_fwd / howl_lstm_bwd
//   nn.Linear(128,256) - ReLU - nn.Linear(256,C) (rnn.py:44-48,71,91) ... howl_head_fwd / howl_head_bwd
// (log_softmax + CTCLoss: howl_ctc_loss, ctc.hip)
//
// Internal layout is batch-major: x (B,T,40) [what the fused frontend writes], gate pre-activations / activations
// (B,T,512), cell c (B,T,128), hidden hseq (B,T+1,128) with hseq[b][0] = h0 and hseq[b][t+1] = h_t, so that both
// "h_t" and "h_{t-1}" are plain strided views for the GEMMs.
//
// The recurrence is the latency-critical part: 38-81 strictly sequential steps, W_hh (256 KB) held in registers for the whole
// launch, h_t in LDS.  Two kernel pairs: FOUR sequences per workgroup on v_mfma_f32_4x4x1 (lstm_fwd4 / lstm_bwd4: eight waves,
// 128 W_hh registers per lane, the MFMA A operand by block broadcast; used up to 8 x CUs sequences per GPU) and SIXTEEN per
// workgroup on v_mfma_f32_16x16x4 (lstm_fwd / lstm_bwd: sixteen waves, gate columns permuted so that each wave holds i,f,g,o of
// the same 8 hidden units) beyond.  The input projection x W_ih^T, the head's first layer and all weight gradients are batched
// GEMMs outside the recurrence (howl_gemm.hip.h); the head's thin second layer is vector work (head_out / head_thin_bwd).
#include <math.h>

#include "howl_common.hip.h"
#include "../../include/howl_hip.h"
#include "howl_gemm.hip.h"
#include "howl_logmel.hip.h"
#include "howl_ctc.hip.h"

namespace {

constexpr int HID = 128;
constexpr int G4 = 4 * HID;          // 512 gate columns, PyTorch order i, f, g, o
constexpr int LSTM_THREADS = 1024;   // 16 waves
constexpr int LSTM_WGRAD_SPLITS = 128;
constexpr int LSTM_MAX_IN = HOWL_MAX_MELS;   // input features (mel bins) the workspace is sized for: 96 (stock NUM_MELS is 80)
constexpr int HS = HID + 4;          // LDS row stride of the h tile (16 rows)
constexpr int DGS = G4 + 4;          // LDS row stride of the dG tile

// Gate nonlinearities on the hardware exp2 / reciprocal (1 ulp each): the cell update sits on the critical path of every
// one of the 38-81 sequential steps, and the library expf / tanhf / division cost ~10x the instructions.  Absolute error
// ~1e-7, far inside the 2e-6 the parity tests hold the hidden states to.
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float tanhf_(float x) {
    // no clamp needed: 2^(+big) = inf -> rcp 0 -> 1; 2^(-big) = 0 -> rcp 1 -> -1 (v_exp_f32 / v_rcp_f32 saturate cleanly)
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * x));
}

// ---------------------------------------------------------------------------------------------------------
// W_hh (512,128) -> register fragments.
//   forward : wave w, tile a in {0,1}, k-step kk: B[k][n] = W_hh[col(w,a,n)][4kk + k],
//             col = (2a + (n >> 3)) * 128 + 8w + (n & 7)   (tile 0 = [i | f], tile 1 = [g | o] of units 8w..8w+7)
//   backward: wave w -> hidden tile nt = w & 7, K half kh = w >> 3, k-step kk (64 per half):
//             B[k][n] = W_hh[256kh + 4kk + k][16nt + n]      (dh = dG . W_hh)
// packed as [wave][frag 0..63][lane]
// ---------------------------------------------------------------------------------------------------------
// (the four-sequence kernels read W_hh in place: their fragments are rows / coalesced columns of the matrix as it is)
// The same launch folds the two bias vectors of the input projection (bsum = b_ih + b_hh).
__global__ void lstm_pack_kernel(const float* __restrict__ whh, float* __restrict__ pf, float* __restrict__ pb,
                                 const float* __restrict__ b_ih, const float* __restrict__ b_hh, float* __restrict__ bsum) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 16 * 64 * 64) return;
    if (idx < G4) bsum[idx] = b_ih[idx] + b_hh[idx];
    const int lane = idx & 63, frag = (idx >> 6) & 63, w = idx >> 12;
    const int k = lane >> 4, n = lane & 15;
    {
        const int a = frag >> 5, kk = frag & 31;
        const int col = (2 * a + (n >> 3)) * HID + 8 * w + (n & 7);
        pf[idx] = whh[col * HID + 4 * kk + k];
    }
    {
        const int nt = w & 7, kh = w >> 3;
        pb[idx] = whh[(256 * kh + 4 * frag + k) * HID + 16 * nt + n];
    }
}

// The A operand of the four-sequence recurrences is the same for all sixteen blocks of v_mfma_f32_4x4x1_16b_f32 (the four
// sequences' h or dG values of one k).  With cbsz = 4 the instruction takes A from the lanes of ONE block (abid) for all
// sixteen, so a lane (block J, row i) loads only the four k values 4J..4J+3 of its row -- one 16-byte LDS read per wave and
// step instead of sixteen (which was as much LDS-pipe time per step, 256 x 1 KB, as the step's matrix work) -- and the
// sixty-four instructions walk abid over the blocks.  acc[e] is the chain of k = e mod 4.
template <int J, int OFF, int N, int JEND = 16>
__device__ __forceinline__ void bcast_mfma64(const float4& a, const float (&w)[N], f32x4 (&acc)[4]) {
    acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.x, w[OFF + 4 * J + 0], acc[0], 4, J, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.y, w[OFF + 4 * J + 1], acc[1], 4, J, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.z, w[OFF + 4 * J + 2], acc[2], 4, J, 0);
    acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.w, w[OFF + 4 * J + 3], acc[3], 4, J, 0);
    if constexpr (J + 1 < JEND) bcast_mfma64<J + 1, OFF, N, JEND>(a, w, acc);
}
// 4 x 4 transpose inside every quad of lanes: v[r] of lane 4q + l  <->  v[l] of lane 4q + r.  Two butterfly stages: the
// partner's registers arrive by v_mov_b32_dpp quad_perm, a select on the lane's parity keeps or takes (16 instructions; a DPP
// bank mask cannot do the select: its banks are whole quads).
__device__ __forceinline__ void quad_transpose(float (&v)[4], bool bit0, bool bit1) {
    auto xor1 = [](float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true)); };
    auto xor2 = [](float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, true)); };
    // stage 1: lane bit 0 <-> register bit 0: even lanes take the partner's v[0] as v[1], odd lanes its v[1] as v[0]
    {
        const float p0 = xor1(v[0]), p1 = xor1(v[1]), p2 = xor1(v[2]), p3 = xor1(v[3]);
        v[0] = bit0 ? p1 : v[0];
        v[1] = bit0 ? v[1] : p0;
        v[2] = bit0 ? p3 : v[2];
        v[3] = bit0 ? v[3] : p2;
    }
    // stage 2: lane bit 1 <-> register bit 1
    {
        const float p0 = xor2(v[0]), p1 = xor2(v[1]), p2 = xor2(v[2]), p3 = xor2(v[3]);
        v[0] = bit1 ? p2 : v[0];
        v[2] = bit1 ? v[2] : p0;
        v[1] = bit1 ? p3 : v[1];
        v[3] = bit1 ? v[3] : p1;
    }
}

// Forward recurrence with FOUR sequences per workgroup on v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products
// per instruction: block = hidden unit, block column = gate, block row = sequence), so the batch dimension is not padded to
// the 16 rows of the 16x16x4 tile: 128 workgroups at B = 512 and a ~1 us matrix floor per step (256 instructions per SIMD).
// Everything around the matrix work is sized to stay out of its way -- measured (tools/lstm_variants.py) the previous
// sixteen-wave version spent 1.15 us of each 2.1 us step outside the MFMAs: four waves per SIMD all running the gate
// arithmetic (with each cell updated in four lanes), two barriers and an LDS exchange of the K halves:
//   eight waves, wave c = units 16c..16c+15 x 4 gates over the WHOLE K (128 W_hh registers per lane, two waves per SIMD):
//   no partial sums to exchange;
//   MFMA role of lane 4j+g: gate g of unit 16c+j for the four sequences (its four accumulator rows); after the gate
//   nonlinearity a 4x4 transpose inside the quad hands lane 4j+g the four gates of (sequence g, unit 16c+j): every cell is
//   updated exactly once;
//   saved activations go straight to HBM from the lanes that hold them (no staging, no second barrier); sequences past the
//   end of the batch are computed as copies of the last one, so that every store is unconditional (identical values) and
//   the compiler can count outstanding memory operations instead of draining them at every step;
//   ONE barrier per step: h_t is double-buffered in LDS.
//   XM > 0 (round 4): the INPUT PROJECTION x_t W_ih^T is part of the step -- XM / 4 more A blocks (the four sequences' x_t, 160
//   floats per step, staged in LDS a step ahead) against XM more weight registers per lane: 4 XM more instructions on the 512 of
//   a step, in exchange for the projection's launch (18 us at 512 x 38 x 40) and its (B, T, 512) buffer written and read back.
//   `gx` is then x itself: row (b, t) at (b * xframes + t) * XM floats.
constexpr int F8_THREADS = 512;
constexpr int F8_HS = HID + 4;          // h rows in LDS (16-byte aligned rows for the float4 A-fragment reads)
constexpr int F8_XS = 64 + 4;           // x rows in LDS (every lane of a wave reads its float4; columns XM.. are never multiplied)
// RIDE (round 5): blocks behind the first `nrec` run the log-mel frontend of the NEXT batch (logmel_body, eight waves) on the CUs
// this recurrence leaves idle -- (B + 3) / 4 workgroups of 38-81 dependent steps: half of the device at B = 512.  Independent work:
// its output is read by later launches only, so there is no flag and nobody waits; the recurrence's blocks come first in dispatch
// order.  (The same frontend as its own launch costs 15.7 us per 512 x 0.5 s step on the critical path.)
template <int XM, bool RIDE = false>
__global__ __launch_bounds__(F8_THREADS) void lstm_fwd4_kernel(const float* __restrict__ gx, const float* __restrict__ wih, int xframes,
                                                               const float* __restrict__ whh,
                                                               const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                               const long long* __restrict__ lengths,
                                                               const float* __restrict__ h0, const float* __restrict__ c0,
                                                               float* __restrict__ gates, float* __restrict__ cs,
                                                               float* __restrict__ hseq, float* __restrict__ hT,
                                                               float* __restrict__ cT, int B, int T, int Tout, int nrec,
                                                               LogmelLaunch next, int next_blocks) {
    if constexpr (RIDE) {
        if ((int)blockIdx.x >= nrec) {
            logmel_body<F8_THREADS / 64, NG_BANDED>(next.pcm, next.L, next.ld, next.T, next.total, next.fbp, next.M, next.log_eps, next.zmuv,
                                                    next.out, next.layout, next.n_quads, next.aligned, blockIdx.x - (unsigned)nrec,
                                                    (unsigned)next_blocks, next.Mo);
            return;
        }
    }
    __shared__ __attribute__((aligned(16))) float hbuf[2][4 * F8_HS];
    __shared__ __attribute__((aligned(16))) float xbuf[2][4 * F8_XS];
    constexpr int XP = XM > 0 ? XM : 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int c = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane >> 2, g = lane & 3;
    const int u = 16 * c + j;                       // hidden unit of this lane's quad
    const int col = g * HID + u;                    // MFMA role: its gate column in PyTorch order (i, f, g, o)
    const int b0 = blockIdx.x * 4;
    // W_hh[col][0..127] straight from the parameter (a 512-byte row per lane, 64 KB per wave out of L2) and the lane's bias
    // b_ih[col] + b_hh[col], which the input projection leaves out for this kernel: no packing launch
    float wB[128];
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(whh + (size_t)col * HID + 4 * q);
        wB[4 * q + 0] = v.x;
        wB[4 * q + 1] = v.y;
        wB[4 * q + 2] = v.z;
        wB[4 * q + 3] = v.w;
    }
    const float bias = b_ih[col] + b_hh[col];
    float wI[XP];
    if constexpr (XM > 0) {
#pragma unroll
        for (int q = 0; q < XM / 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(wih + (size_t)col * XM + 4 * q);
            wI[4 * q + 0] = v.x;
            wI[4 * q + 1] = v.y;
            wI[4 * q + 2] = v.z;
            wI[4 * q + 3] = v.w;
        }
    }
    // cell role: sequence g of the workgroup, unit u
    const int bc = min(b0 + g, B - 1);
    float cst = c0 != nullptr ? c0[(size_t)bc * HID + u] : 0.0f;
    float hst = h0 != nullptr ? h0[(size_t)bc * HID + u] : 0.0f;
    const int len = lengths != nullptr ? (int)lengths[bc] : T;
    hbuf[0][g * F8_HS + u] = hst;
    hseq[((size_t)bc * (T + 1)) * HID + u] = hst;
    // Running BYTE offsets from the (uniform) buffer bases, advanced once per step: the accesses become
    // global_load/store v, v_offset, s[base] with nothing else to compute (element indices cost an add and a 64-bit
    // shift-add per access and step: 28 VALU instructions of a ~130-instruction step).  < 4 GB by the launcher's check.
    auto ldg = [](const float* base, unsigned boff) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + boff); };
    auto stg = [](float* base, unsigned boff, float v) { *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + boff) = v; };
    unsigned cb = ((unsigned)bc * (unsigned)T * HID + u) * 4u;                  // cs[bc][t][u]
    unsigned hb = (((unsigned)bc * (unsigned)(T + 1) + 1u) * HID + u) * 4u;     // hseq[bc][t + 1][u]
    // MFMA role: (sequence r, step t, column col) in gx and gates (same shape)
    unsigned gb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) gb[r] = ((unsigned)min(b0 + r, B - 1) * (unsigned)T * G4 + col) * 4u;
    float nx[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    // XM > 0: thread (sequence xr, feature xm) of the first 4 XM carries x_t of the NEXT step into xbuf (requested a step ahead)
    const int xr = min(tid / XP, 3), xm = tid - (tid / XP) * XP;
    const bool xthread = XM > 0 && tid < 4 * XM;
    unsigned xb = ((unsigned)min(b0 + xr, B - 1) * (unsigned)xframes * XP + xm) * 4u;      // x[b][t][m]
    float xn = 0.0f;
    if constexpr (XM > 0) {
        if (xthread) xbuf[0][xr * F8_XS + xm] = ldg(gx, xb);
        xb += Tout > 1 ? XP * 4u : 0u;
        xn = ldg(gx, xb);         // x_1
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) nx[r] = ldg(gx, gb[r]);
    }
    // sigmoid(x) = 1 / (1 + 2^(-x log2 e)); the cell-candidate gate is tanh(x) = 2 sigmoid(2x) - 1
    const float kneg = g == 2 ? -2.88539008177792681f : -1.44269504088896341f;
    const float amul = g == 2 ? 2.0f : 1.0f, aadd = g == 2 ? -1.0f : 0.0f;
    __syncthreads();
    // the 128 fragment loads are complete before the loop (otherwise the first trip's bookkeeping, one s_waitcnt per
    // fragment, stays in the loop body for every step)
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0) on gfx9
    for (int t = 0; t < Tout; ++t) {
        const float* hcur = hbuf[t & 1];
        float* hnxt = hbuf[(t + 1) & 1];
        float pre[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) pre[r] = nx[r] + bias;
        float xw = xn;      // x_{t+1}, requested a step ago; goes to LDS at the end of this step
        if constexpr (XM > 0) {   // x_{t+2} (the last steps re-read the last frame: an unconditional load keeps the count exact)
            xb += t + 2 < Tout ? XP * 4u : 0u;
            xn = ldg(gx, xb);
        } else {   // next step's input-projection terms (the last step re-reads its own)
            const float* gxn = gx + (t + 1 < Tout ? G4 : 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) nx[r] = ldg(gxn, gb[r]);
        }
        // A_j[i] comes from lane 4j+i: row i = lane & 3 = g; this lane's k values are 4j .. 4j+3 and 64 + 4j .. 64 + 4j+3
        const float4 alo = *reinterpret_cast<const float4*>(hcur + g * F8_HS + 4 * j);
        const float4 ahi = *reinterpret_cast<const float4*>(hcur + g * F8_HS + 64 + 4 * j);
        f32x4 acc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = {0.0f, 0.0f, 0.0f, 0.0f};
        // (the input's share at the END of the previous step, in the shadow of its barrier -- it does not depend on h -- was
        // measured SLOWER: 86.0 against 79.5 us per 38-step launch)
        if constexpr (XM > 0) {
            const float4 ax = *reinterpret_cast<const float4*>(&xbuf[t & 1][g * F8_XS + 4 * j]);
            bcast_mfma64<0, 0, XP, XM / 4>(ax, wI, acc);
        }
        bcast_mfma64<0, 0>(alo, wB, acc);
        bcast_mfma64<0, 64>(ahi, wB, acc);
        const f32x4 sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        float act[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = sum[r] + pre[r];
            const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(kneg * p));
            act[r] = fmaf(sg, amul, aadd);
            stg(gates, gb[r], act[r]);
            gb[r] += G4 * 4u;
        }
        quad_transpose(act, (g & 1) != 0, (g & 2) != 0);        // -> i, f, g, o of (sequence g, unit u)
        const bool live = t < len;
        const float cn = act[1] * cst + act[0] * act[2];
        const float hn = act[3] * tanhf_(cn);
        cst = live ? cn : cst;
        hst = live ? hn : hst;
        hnxt[g * F8_HS + u] = hst;
        if (xthread) xbuf[(t + 1) & 1][xr * F8_XS + xm] = xw;
        stg(cs, cb, cn);
        stg(hseq, hb, live ? hn : 0.0f);      // padded outputs are zero
        cb += HID * 4u;
        hb += HID * 4u;
        __syncthreads();
    }
    hT[(size_t)bc * HID + u] = hst;
    cT[(size_t)bc * HID + u] = cst;
}

// forward recurrence.  gx: (B,T,512) = x W_ih^T + b_ih + b_hh.  Outputs gates (B,T,512) post-activation, c (B,T,128),
// hseq (B,T+1,128) [row 0 = h0], and the final (h, c).  lengths == nullptr: every sequence runs T steps.
__global__ __launch_bounds__(LSTM_THREADS) void lstm_fwd_kernel(const float* __restrict__ gx, const float* __restrict__ pf,
                                                                const long long* __restrict__ lengths,
                                                                const float* __restrict__ h0, const float* __restrict__ c0,
                                                                float* __restrict__ gates, float* __restrict__ cs,
                                                                float* __restrict__ hseq, float* __restrict__ hT,
                                                                float* __restrict__ cT, int B, int T, int Tout) {
    HIP_DYNAMIC_SHARED(float, lds_fwd)   // 115 KB: above the static limit
    float (*hbuf)[16 * HS] = reinterpret_cast<float (*)[16 * HS]>(lds_fwd);
    // Saved activations of a step (gates i,f,g,o | c | h: 768 floats per sequence) are collected in LDS and written out by
    // all 1024 threads as three 16-byte stores each during the NEXT step's MFMAs.  Written straight from the owner lanes
    // they were 24 scattered 4-byte stores per lane and step -- ~400 store instructions per CU and step, more
    // vector-memory issue time than the step's matrix work.
    constexpr int SROW = 6 * HID;
    float (*sbuf)[16 * SROW] = reinterpret_cast<float (*)[16 * SROW]>(lds_fwd + 2 * 16 * HS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = blockIdx.x * 16;
    // Destination of this thread's three float4 slots: a 32-bit element offset at t = 0 plus a per-step stride, packed so
    // that nothing 64-bit has to stay live across the steps (the compiler used to hoist six 64-bit bases, spill them at the
    // 128-VGPR limit of a 16-wave workgroup and reload each one from scratch -- with a full vmcnt(0) wait -- in front of
    // every store of every step: ~2 us of each 6.7 us step).  kind: 0 gates, 1 cs, 2 hseq, 3 nothing.
    unsigned foff[3];
    int fkind[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int i4 = tid + i * LSTM_THREADS;          // float4 index: 16 rows x 192
        const int row = i4 / (SROW / 4), c4 = i4 - row * (SROW / 4);
        const int b = b0 + row;
        if (b >= B) {
            fkind[i] = 3;
            foff[i] = 0u;
        } else if (c4 < G4 / 4) {
            fkind[i] = 0;
            foff[i] = (unsigned)b * (unsigned)T * G4 + 4u * c4;
        } else if (c4 < (G4 + HID) / 4) {
            fkind[i] = 1;
            foff[i] = (unsigned)b * (unsigned)T * HID + 4u * (c4 - G4 / 4);
        } else {
            fkind[i] = 2;
            foff[i] = ((unsigned)b * (unsigned)(T + 1) + 1u) * HID + 4u * (c4 - (G4 + HID) / 4);
        }
    }
    auto flush = [&](int t, const float* sb) {   // step t's rows -> gates / cs / hseq
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int i4 = tid + i * LSTM_THREADS;
            const float4 v = *reinterpret_cast<const float4*>(sb + 4 * i4);   // row * SROW + 4 * c4 == 4 * i4
            const unsigned o = foff[i] + (unsigned)t * (fkind[i] == 0 ? (unsigned)G4 : (unsigned)HID);
            // three predicated stores off uniform base pointers (a selected pointer would go through a table in scratch)
            if (fkind[i] == 0) *reinterpret_cast<float4*>(gates + o) = v;
            else if (fkind[i] == 1) *reinterpret_cast<float4*>(cs + o) = v;
            else if (fkind[i] == 2) *reinterpret_cast<float4*>(hseq + o) = v;
        }
    };
    float wA[32], wB[32];
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) {
        wA[kk] = pf[((size_t)wave * 64 + kk) * 64 + lane];
        wB[kk] = pf[((size_t)wave * 64 + 32 + kk) * 64 + lane];
    }
    const int n = lane & 15, u = 8 * wave + (n & 7);
    // Lane pair (l, l^8) holds the i|f and g|o pre-activations of unit u for rows 4*(lane>>4) + {0..3}.  Both lanes work
    // on the cell update: the lane with n < 8 takes rows +0,+1, its partner rows +2,+3 (they swap the two gate values the
    // other one needs), so the transcendental-heavy update costs half the instructions of an owner-lane-only scheme.
    const bool hi = n >= 8;
    float cst[2], hst[2];
    int len[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = 4 * (lane >> 4) + (hi ? 2 : 0) + q;
        const int b = b0 + row;
        const bool vb = b < B;
        cst[q] = (vb && c0 != nullptr) ? c0[(size_t)b * HID + u] : 0.0f;
        hst[q] = (vb && h0 != nullptr) ? h0[(size_t)b * HID + u] : 0.0f;
        len[q] = vb ? (lengths != nullptr ? (int)lengths[b] : T) : 0;
        hbuf[0][row * HS + u] = hst[q];
        if (vb) hseq[((size_t)b * (T + 1)) * HID + u] = hst[q];
    }
    __syncthreads();
    const int colA = (n >> 3) * HID + u, colB = (2 + (n >> 3)) * HID + u;   // this lane's columns in tiles [i|f], [g|o]
    // input-projection terms of step t+1 are fetched while step t multiplies (a dependent global load per step would
    // put HBM latency on the critical path of every one of the 38-81 sequential steps)
    f32x4 nxA, nxB;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = b0 + 4 * (lane >> 4) + r;
        const size_t go = ((size_t)b * T) * G4;
        nxA[r] = b < B ? gx[go + colA] : 0.0f;
        nxB[r] = b < B ? gx[go + colB] : 0.0f;
    }
    for (int t = 0; t < Tout; ++t) {
        const float* hcur = hbuf[t & 1];
        float* hnxt = hbuf[(t + 1) & 1];
        float* scur = sbuf[t & 1];
        if (t > 0) flush(t - 1, sbuf[(t - 1) & 1]);   // complete and visible since the barrier that ended step t-1
        f32x4 accA = nxA, accB = nxB;
        if (t + 1 < Tout) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = b0 + 4 * (lane >> 4) + r;
                const size_t go = ((size_t)b * T + t + 1) * G4;
                nxA[r] = b < B ? gx[go + colA] : 0.0f;
                nxB[r] = b < B ? gx[go + colB] : 0.0f;
            }
        }
        const float* arow = hcur + (lane & 15) * HS + (lane >> 4);
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            const float a = arow[4 * kk];
            accA = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wA[kk], accA, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wB[kk], accB, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            // this lane updates row +(2 hi + q): it holds (i,g) [low lane] or (f,o) [high lane] of that row and receives
            // the other pair from its partner, to which it sends its values of the partner's row
            const float ownA = hi ? accA[2 + q] : accA[q], ownB = hi ? accB[2 + q] : accB[q];
            const float sendA = hi ? accA[q] : accA[2 + q], sendB = hi ? accB[q] : accB[2 + q];
            const float recvA = __shfl_xor(sendA, 8), recvB = __shfl_xor(sendB, 8);
            const int row = 4 * (lane >> 4) + (hi ? 2 : 0) + q;
            const bool live = t < len[q];
            const float ig = sigmoidf_(hi ? recvA : ownA), fg = sigmoidf_(hi ? ownA : recvA);
            const float gg = tanhf_(hi ? recvB : ownB), og = sigmoidf_(hi ? ownB : recvB);
            const float cn = fg * cst[q] + ig * gg;
            const float hn = og * tanhf_(cn);
            if (live) {
                cst[q] = cn;
                hst[q] = hn;
            }
            hnxt[row * HS + u] = hst[q];
            float* sr = scur + row * SROW + u;
            sr[0] = ig;
            sr[HID] = fg;
            sr[2 * HID] = gg;
            sr[3 * HID] = og;
            sr[4 * HID] = cn;
            sr[5 * HID] = live ? hn : 0.0f;   // padded outputs are zero
        }
        __syncthreads();
    }
    if (Tout > 0) flush(Tout - 1, sbuf[(Tout - 1) & 1]);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int b = b0 + 4 * (lane >> 4) + (hi ? 2 : 0) + q;
        if (b < B) {
            hT[(size_t)b * HID + u] = hst[q];
            cT[(size_t)b * HID + u] = cst[q];
        }
    }
}

// BPTT recurrence.  dy: (B,T,128) gradient w.r.t. the padded outputs h_t (nullptr: none); dhT/dcT: gradient w.r.t. the
// final states (nullptr: none).  Writes dG (B,T,512), the pre-activation gate gradients (zero beyond each length),
// for the batched weight-gradient GEMMs.
__global__ __launch_bounds__(LSTM_THREADS) void lstm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ dhT,
                                                                const float* __restrict__ dcT, const float* __restrict__ pb,
                                                                const long long* __restrict__ lengths,
                                                                const float* __restrict__ gates, const float* __restrict__ cs,
                                                                const float* __restrict__ c0, float* __restrict__ dG, int B,
                                                                int T, int Tout) {
    __shared__ float dgt[16 * DGS];          // this step's dG tile (MFMA A operand)
    __shared__ float part[2][16 * HS];       // the two K halves of dh_{t-1} = dG_t . W_hh
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = blockIdx.x * 16;
    float wk[64];
#pragma unroll
    for (int kk = 0; kk < 64; ++kk) wk[kk] = pb[((size_t)wave * 64 + kk) * 64 + lane];
    const int nt = wave & 7, kh = wave >> 3;
    // elementwise phase: thread owns cells (row = tid >> 6, unit = tid & 63) and (row, unit + 64)
    const int row = tid >> 6, b = b0 + row;
    const bool vb = b < B;
    const int len = vb ? (lengths != nullptr ? (int)lengths[b] : T) : 0;
    float dc[2], dhp[2];   // running dL/dc_t and the pass-through part of dL/dh_t
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int uu = (tid & 63) + 64 * q;
        dc[q] = (vb && dcT != nullptr) ? dcT[(size_t)b * HID + uu] : 0.0f;
        dhp[q] = (vb && dhT != nullptr) ? dhT[(size_t)b * HID + uu] : 0.0f;
        part[0][row * HS + uu] = 0.0f;
        part[1][row * HS + uu] = 0.0f;
    }
    __syncthreads();
    // saved activations of the current step, fetched one step ahead (see lstm_fwd_kernel)
    struct Step {
        float ig, fg, gg, og, cn, cp, dyv;
    };
    // Unconditional loads from clamped addresses, zeroed afterwards: predicated loads (and the two alternative sources of
    // c_{t-1}) were compiled into groups separated by vmcnt(0) waits, i.e. several HBM round trips per step.
    const size_t bclamp = vb ? (size_t)b : 0;
    float c0v[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) c0v[q] = (vb && c0 != nullptr) ? c0[(size_t)b * HID + (tid & 63) + 64 * q] : 0.0f;
    auto fetch = [&](int t, int q, Step& st) {
        const int uu = (tid & 63) + 64 * q;
        const bool lv = vb && t >= 0 && t < len;
        const size_t o = bclamp * T + (lv ? t : 0);
        const size_t op = o - ((lv && t > 0) ? 1 : 0);
        const float ig = gates[o * G4 + uu], fg = gates[o * G4 + HID + uu];
        const float gg = gates[o * G4 + 2 * HID + uu], og = gates[o * G4 + 3 * HID + uu];
        const float cn = cs[o * HID + uu], cpv = cs[op * HID + uu];
        float dyv = 0.0f;
        if (dy != nullptr) dyv = dy[o * HID + uu];   // uniform branch
        st.ig = lv ? ig : 0.0f;
        st.fg = lv ? fg : 0.0f;
        st.gg = lv ? gg : 0.0f;
        st.og = lv ? og : 0.0f;
        st.cn = lv ? cn : 0.0f;
        st.cp = lv ? (t > 0 ? cpv : c0v[q]) : 0.0f;
        st.dyv = lv ? dyv : 0.0f;
    };
    Step cur[2], nxt[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) fetch(Tout - 1, q, cur[q]);
    for (int t = Tout - 1; t >= 0; --t) {
        const bool live = t < len;
#pragma unroll
        for (int q = 0; q < 2; ++q) fetch(t - 1, q, nxt[q]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int uu = (tid & 63) + 64 * q;
            // dL/dh_t = (recurrent term from step t+1) + (pass-through when h was frozen) + (output gradient)
            float dh = part[0][row * HS + uu] + part[1][row * HS + uu] + dhp[q];
            float di = 0.0f, df = 0.0f, dg = 0.0f, dov = 0.0f;
            if (live) {
                const Step& st = cur[q];
                dh += st.dyv;
                const float tc = tanhf_(st.cn);
                dov = dh * tc * st.og * (1.0f - st.og);
                const float dct = dc[q] + dh * st.og * (1.0f - tc * tc);
                di = dct * st.gg * st.ig * (1.0f - st.ig);
                dg = dct * st.ig * (1.0f - st.gg * st.gg);
                df = dct * st.cp * st.fg * (1.0f - st.fg);
                dc[q] = dct * st.fg;
                dhp[q] = 0.0f;      // h_t was produced by the cell: everything flows through the gates
            } else {
                dhp[q] = dh;        // frozen step (t >= length): h_t = h_{t-1}, c_t = c_{t-1}
            }
            dgt[row * DGS + uu] = di;
            dgt[row * DGS + HID + uu] = df;
            dgt[row * DGS + 2 * HID + uu] = dg;
            dgt[row * DGS + 3 * HID + uu] = dov;
            if (vb) {
                const size_t o = ((size_t)b * T + t) * G4;
                dG[o + uu] = di;
                dG[o + HID + uu] = df;
                dG[o + 2 * HID + uu] = dg;
                dG[o + 3 * HID + uu] = dov;
            }
            cur[q] = nxt[q];
        }
        __syncthreads();
        // dh_{t-1}[row][16nt + n] (K half kh) = sum_col dG[row][256kh + col] * W_hh[256kh + col][16nt + n]
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        const float* arow = dgt + (lane & 15) * DGS + 256 * kh + (lane >> 4);
#pragma unroll
        for (int kk = 0; kk < 64; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * kk], wk[kk], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) part[kh][(4 * (lane >> 4) + r) * HS + 16 * nt + (lane & 15)] = acc[r];
        __syncthreads();
    }
}

// Backward recurrence with four sequences per workgroup (see lstm_fwd4_kernel): dh_{t-1}[4][128] = dG_t[4][512] . W_hh on
// v_mfma_f32_4x4x1_16b_f32.  Eight waves: wave w = (hidden-column chunk ch = w & 1, gate-column range kr = w >> 1: 128 of the
// 512 K values, 128 W_hh registers per lane); the four K ranges meet in LDS and are summed in a fixed order by the thread that
// owns the cell: thread (sequence tid >> 7, unit tid & 127) -- every thread has one.
constexpr int B4_DGS = G4 + 4;          // dG rows in LDS (16-byte aligned rows for the float4 A-fragment reads)
constexpr int B4_PS = HID + 4;
constexpr int B8_THREADS = 512;
__global__ __launch_bounds__(B8_THREADS) void lstm_bwd4_kernel(const float* __restrict__ dy, const float* __restrict__ dhT,
                                                                 const float* __restrict__ dcT, const float* __restrict__ whh,
                                                                 const long long* __restrict__ lengths,
                                                                 const float* __restrict__ gates, const float* __restrict__ cs,
                                                                 const float* __restrict__ c0, float* __restrict__ dG,
                                                                 float* __restrict__ bpart, int B, int T, int Tout, int nrec,
                                                                 WgradJob ride) {
    // Blocks behind the first `nrec` do an unrelated job in the same launch: a weight gradient that does not depend on this
    // recurrence (the head's first layer), on the CUs the recurrence leaves idle -- (B + 3) / 4 workgroups of 38-81 dependent
    // steps occupy half of the device at B = 512.  The recurrence's blocks come first in dispatch order and wait for nobody.
    if ((int)blockIdx.x >= nrec) {
        __shared__ __attribute__((aligned(16))) float As[2][WG_K * WG_LDA];
        __shared__ __attribute__((aligned(16))) float Bs[2][WG_K * WG_LDA];
        wgrad_w8_body(ride, (int)blockIdx.x - nrec, As, Bs);
        return;
    }
    __shared__ __attribute__((aligned(16))) float dgt[4 * B4_DGS];   // this step's dG rows (MFMA A operand)
    __shared__ float partd[4][4 * B4_PS];                             // the four K ranges of dh_{t-1}
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ch = wave & 1, kr = wave >> 1;
    const int b0 = blockIdx.x * 4;
    float wk[128];      // W_hh[128 kr + k][64 ch + lane], read in place (coalesced over the lanes)
#pragma unroll
    for (int kk = 0; kk < 128; ++kk) wk[kk] = whh[(size_t)(128 * kr + kk) * HID + 64 * ch + lane];
    // elementwise part: one cell per thread.  Sequences past the end of the batch are computed as copies of the last one
    // (identical loads, identical stores, left out of the bias sums): no store of the step loop is conditional.
    const int s = tid >> 7, u = tid & (HID - 1);
    const int b = min(b0 + s, B - 1);
    const bool dup = b0 + s >= B;
    const int len = lengths != nullptr ? (int)lengths[b] : T;
    float dc = dcT != nullptr ? dcT[(size_t)b * HID + u] : 0.0f;     // running dL/dc_t
    float dhp = dhT != nullptr ? dhT[(size_t)b * HID + u] : 0.0f;    // pass-through part of dL/dh_t
    float bs_i = 0.0f, bs_f = 0.0f, bs_g = 0.0f, bs_o = 0.0f;       // this cell's dG summed over the steps (bias gradient)
    for (int i = tid; i < 4 * 4 * B4_PS; i += B8_THREADS) (&partd[0][0])[i] = 0.0f;
    struct Step {
        float ig, fg, gg, og, cn, cp, dyv;
    };
    const float c0v = c0 != nullptr ? c0[(size_t)b * HID + u] : 0.0f;
    // A step's saved activations are REQUESTED right after the previous step's gate gradients went out and TAKEN (masked)
    // after the matrix work of that step: a full step of latency cover.  Taken where they were requested -- as the
    // compiler scheduled it when left alone -- every step waited ~0.35 us for HBM (tools/lstm_variants.py: noload).
    Step raw;
    auto request = [&](int t) {   // unconditional loads from clamped addresses
        const bool lv = t >= 0 && t < len;
        const size_t o = (size_t)b * T + (lv ? t : 0);
        const size_t op = o - ((lv && t > 0) ? 1 : 0);
        raw.ig = gates[o * G4 + u];
        raw.fg = gates[o * G4 + HID + u];
        raw.gg = gates[o * G4 + 2 * HID + u];
        raw.og = gates[o * G4 + 3 * HID + u];
        raw.cn = cs[o * HID + u];
        raw.cp = cs[op * HID + u];
        raw.dyv = dy != nullptr ? dy[o * HID + u] : 0.0f;
    };
    auto take = [&](int t, Step& st) {
        const bool lv = t >= 0 && t < len;
        st.ig = lv ? raw.ig : 0.0f;
        st.fg = lv ? raw.fg : 0.0f;
        st.gg = lv ? raw.gg : 0.0f;
        st.og = lv ? raw.og : 0.0f;
        st.cn = lv ? raw.cn : 0.0f;
        st.cp = lv ? (t > 0 ? raw.cp : c0v) : 0.0f;
        st.dyv = lv ? raw.dyv : 0.0f;
    };
    Step cur;
    request(Tout - 1);
    take(Tout - 1, cur);
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): the fragment loads are not the loop's business (see lstm_fwd4_kernel)
    for (int t = Tout - 1; t >= 0; --t) {
        {
            // dL/dh_t = (recurrent term from step t+1) + (pass-through when h was frozen) + (output gradient)
            float dh = dhp;
#pragma unroll
            for (int r = 0; r < 4; ++r) dh += partd[r][s * B4_PS + u];
            float di = 0.0f, df = 0.0f, dg = 0.0f, dov = 0.0f;
            if (t < len) {
                dh += cur.dyv;
                const float tc = tanhf_(cur.cn);
                dov = dh * tc * cur.og * (1.0f - cur.og);
                const float dct = dc + dh * cur.og * (1.0f - tc * tc);
                di = dct * cur.gg * cur.ig * (1.0f - cur.ig);
                dg = dct * cur.ig * (1.0f - cur.gg * cur.gg);
                df = dct * cur.cp * cur.fg * (1.0f - cur.fg);
                dc = dct * cur.fg;
                dhp = 0.0f;      // h_t was produced by the cell: everything flows through the gates
            } else {
                dhp = dh;        // frozen step (t >= length): h_t = h_{t-1}, c_t = c_{t-1}
            }
            float* dr = dgt + s * B4_DGS + u;
            dr[0] = di;
            dr[HID] = df;
            dr[2 * HID] = dg;
            dr[3 * HID] = dov;
            bs_i += di;
            bs_f += df;
            bs_g += dg;
            bs_o += dov;
            float* go = dG + ((size_t)b * T + t) * G4 + u;
            go[0] = di;
            go[HID] = df;
            go[2 * HID] = dg;
            go[3 * HID] = dov;
            request(t - 1);
        }
        __syncthreads();
        // dh_{t-1}[seq][64 ch + lane] (K range kr) = sum_k dG[seq][128 kr + k] * W_hh[128 kr + k][64 ch + lane]
        f32x4 acc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = {0.0f, 0.0f, 0.0f, 0.0f};
        // A_j[i] comes from lane 4j+i: this lane's k values are gate columns 128 kr + 4j .. 4j+3 and + 64 of sequence lane & 3
        const float* arow = dgt + (lane & 3) * B4_DGS + 128 * kr + 4 * (lane >> 2);
        const float4 alo = *reinterpret_cast<const float4*>(arow), ahi = *reinterpret_cast<const float4*>(arow + 64);
        bcast_mfma64<0, 0>(alo, wk, acc);
        bcast_mfma64<0, 64>(ahi, wk, acc);
        const f32x4 sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
        for (int r = 0; r < 4; ++r) partd[kr][r * B4_PS + 64 * ch + lane] = sum[r];
        __builtin_amdgcn_sched_barrier(0);     // the saved activations are not touched before this point
        take(t - 1, cur);
        __syncthreads();
    }
    if (dup) bs_i = bs_f = bs_g = bs_o = 0.0f;
    // bias gradient: column sums of dG over (sequence, step) -- the step sums are in registers, the four sequences meet in
    // the dG tile; bpart[workgroup][512] is folded with the weight gradients' split-K slabs (no pass over dG in HBM)
    {
        float* dr = dgt + s * B4_DGS + u;
        dr[0] = bs_i;
        dr[HID] = bs_f;
        dr[2 * HID] = bs_g;
        dr[3 * HID] = bs_o;
    }
    __syncthreads();
    if (tid < G4)
        bpart[(size_t)blockIdx.x * G4 + tid] = ((dgt[tid] + dgt[B4_DGS + tid]) + dgt[2 * B4_DGS + tid]) + dgt[3 * B4_DGS + tid];
}

// ---------------------------------------------------------------------------------------------------------
// The classifier head of both LSTM models: Linear(n_in, n_hid) - ReLU - Linear(n_hid, n_out)  (rnn.py:44-48).
// With the handful of labels these models have (n_out <= 8) the second layer is HBM-bound vector work on the hidden
// activations y1 (rows x n_hid), not a matrix-core problem: a 64-wide MFMA tile would multiply 59 columns of padding.
//   forward : head_out_kernel    y2 = y1 W2^T + b2, sixteen lanes per row
//   backward: head_thin_bwd_kernel, ONE pass over y1 that produces everything of the second layer and the ReLU:
//             dz1 = (y1 > 0) * (dy2 W2)  [stored, feeds the two first-layer GEMMs],  dW2 = dy2^T y1,  db2 = colsum dy2,
//             db1 = colsum dz1  -- per-workgroup partials, folded with the first layer's split-K slabs in one launch.
// (replaces gemm + thin_wgrad + colsum + relu_bwd + colsum: 5 launches and three more passes over rows x n_hid)
// ---------------------------------------------------------------------------------------------------------
constexpr int HEAD_MAX_HID = 256;      // the vector kernels keep a 16-float (forward) / 8-float (backward) strip per lane
constexpr int HEAD_BWD_BLOCKS = 512;
constexpr int HEAD_W1_SPLITS = 128;     // split-K slabs of the first layer's weight gradient (one 128-row tile per CU)   // row chunks of the backward pass = slabs of its partial sums

// sum over the 16 lanes of a DPP row, result in every lane
__device__ __forceinline__ float row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));   // row_mirror
    return v;
}

// lane = (row slot rs = lane >> 4, strip c = lane & 15): columns 4 (c + 16 i) .. +3, i = 0..3 (coalesced 256-byte runs)
template <int NO>
__global__ __launch_bounds__(256) void head_out_kernel(const float* __restrict__ y1, int rows, int n_hid,
                                                       const float* __restrict__ w2, const float* __restrict__ b2,
                                                       float* __restrict__ y2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rs = lane >> 4, c = lane & 15;
    float4 w[NO][4];
    bool okc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        okc[i] = 4 * (c + 16 * i) < n_hid;
#pragma unroll
        for (int n = 0; n < NO; ++n)
            w[n][i] = okc[i] ? *reinterpret_cast<const float4*>(w2 + (size_t)n * n_hid + 4 * (c + 16 * i)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float bias = c < NO ? b2[c] : 0.0f;
    const int stride = gridDim.x * 16;
    for (int r0 = (blockIdx.x * 4 + wave) * 4; r0 < rows; r0 += stride) {
        const int r = r0 + rs;
        const bool okr = r < rows;
        const float* src = y1 + (size_t)(okr ? r : rows - 1) * n_hid;
        float4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = okc[i] ? *reinterpret_cast<const float4*>(src + 4 * (c + 16 * i)) : make_float4(0.f, 0.f, 0.f, 0.f);
        float out = 0.0f;
#pragma unroll
        for (int n = 0; n < NO; ++n) {
            float sn = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                sn += (v[i].x * w[n][i].x + v[i].y * w[n][i].y) + (v[i].z * w[n][i].z + v[i].w * w[n][i].w);
            sn = row16_sum(sn);
            out = c == n ? sn : out;
        }
        if (okr && c < NO) y2[(size_t)r * NO + c] = out + bias;
    }
}

// lane = (row slot rs = lane >> 5, strip c = lane & 31): columns 4 (c + 32 i) .. +3, i = 0, 1; a workgroup walks its row
// chunk eight rows at a time.  part: [block][NO * n_hid (dW2) | n_hid (db1) | NO (db2)]
// loss = mean_b nll_b / max(L_b, 1) (CTCLoss reduction "mean"), in ctc_mean_kernel's summation order (256 threads)
__device__ __forceinline__ void ctc_mean_block(const float* __restrict__ nll, const long long* __restrict__ target_lengths, int B,
                                               float* __restrict__ loss) {
    ctc_mean_256(nll, target_lengths, B, loss);      // (howl_gemm.hip.h: shared with the slab fold's rider)
}
__global__ __launch_bounds__(256) void ctc_mean_only_kernel(HowlCtcMean m) { ctc_mean_block(m.nll, m.target_lengths, m.B, m.loss); }

template <int NO>
__global__ __launch_bounds__(256) void head_thin_bwd_kernel(const float* __restrict__ y1, const float* __restrict__ dy2,
                                                            int rows, int n_hid, int rows_per_block,
                                                            const float* __restrict__ w2, float* __restrict__ dz1,
                                                            float* __restrict__ part, int nblocks, HowlCtcMean cm) {
    if ((int)blockIdx.x == nblocks) {      // one extra block: the batch mean of a CTC loss whose launch was left out
        ctc_mean_block(cm.nll, cm.target_lengths, cm.B, cm.loss);
        return;
    }
    __shared__ float red[4][NO * 8 + 8 + NO][32];      // [wave][value][strip]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rs = lane >> 5, c = lane & 31;
    float4 w[NO][2], aw[NO][2], ab[2];
    float ad[NO];
    bool okc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        okc[i] = 4 * (c + 32 * i) < n_hid;
        ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int n = 0; n < NO; ++n) {
            w[n][i] = okc[i] ? *reinterpret_cast<const float4*>(w2 + (size_t)n * n_hid + 4 * (c + 32 * i)) : make_float4(0.f, 0.f, 0.f, 0.f);
            aw[n][i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int n = 0; n < NO; ++n) ad[n] = 0.0f;
    const int rbeg = blockIdx.x * rows_per_block, rend = min(rows, rbeg + rows_per_block);
    for (int r0 = rbeg + 2 * wave; r0 < rend; r0 += 8) {
        const int r = r0 + rs;
        const bool okr = r < rend;
        const size_t rc = (size_t)(okr ? r : rend - 1);
        float d[NO];
#pragma unroll
        for (int n = 0; n < NO; ++n) {
            const float t = dy2[rc * NO + n];
            d[n] = okr ? t : 0.0f;
            ad[n] += d[n];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (!okc[i]) continue;
            const float4 v = *reinterpret_cast<const float4*>(y1 + rc * n_hid + 4 * (c + 32 * i));
            float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int n = 0; n < NO; ++n) {
                z.x = fmaf(d[n], w[n][i].x, z.x);
                z.y = fmaf(d[n], w[n][i].y, z.y);
                z.z = fmaf(d[n], w[n][i].z, z.z);
                z.w = fmaf(d[n], w[n][i].w, z.w);
                aw[n][i].x = fmaf(d[n], v.x, aw[n][i].x);
                aw[n][i].y = fmaf(d[n], v.y, aw[n][i].y);
                aw[n][i].z = fmaf(d[n], v.z, aw[n][i].z);
                aw[n][i].w = fmaf(d[n], v.w, aw[n][i].w);
            }
            z.x = v.x > 0.0f ? z.x : 0.0f;
            z.y = v.y > 0.0f ? z.y : 0.0f;
            z.z = v.z > 0.0f ? z.z : 0.0f;
            z.w = v.w > 0.0f ? z.w : 0.0f;
            ab[i].x += z.x;
            ab[i].y += z.y;
            ab[i].z += z.z;
            ab[i].w += z.w;
            if (okr) *reinterpret_cast<float4*>(dz1 + rc * n_hid + 4 * (c + 32 * i)) = z;
        }
    }
    // fold the eight row slots of the workgroup in a fixed order: the two of a wave in registers, the four waves through LDS
    auto pair = [&](float v) { return v + __shfl_xor(v, 32); };
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int n = 0; n < NO; ++n) {
            const float4 t = aw[n][i];
            const float a0 = pair(t.x), a1 = pair(t.y), a2 = pair(t.z), a3 = pair(t.w);
            if (rs == 0) {
                red[wave][(n * 2 + i) * 4 + 0][c] = a0;
                red[wave][(n * 2 + i) * 4 + 1][c] = a1;
                red[wave][(n * 2 + i) * 4 + 2][c] = a2;
                red[wave][(n * 2 + i) * 4 + 3][c] = a3;
            }
        }
        const float b0 = pair(ab[i].x), b1 = pair(ab[i].y), b2 = pair(ab[i].z), b3 = pair(ab[i].w);
        if (rs == 0) {
            red[wave][NO * 8 + i * 4 + 0][c] = b0;
            red[wave][NO * 8 + i * 4 + 1][c] = b1;
            red[wave][NO * 8 + i * 4 + 2][c] = b2;
            red[wave][NO * 8 + i * 4 + 3][c] = b3;
        }
    }
#pragma unroll
    for (int n = 0; n < NO; ++n) {
        const float t = pair(ad[n]);     // identical in the 32 strips of a row slot
        if (rs == 0) red[wave][NO * 8 + 8 + n][c] = t;
    }
    __syncthreads();
    float* pb = part + (size_t)blockIdx.x * ((size_t)(NO + 1) * n_hid + NO);
    constexpr int NV = NO * 8 + 8;
    for (int idx = threadIdx.x; idx < NV * 32; idx += 256) {
        const int val = idx >> 5, cc = idx & 31;
        const float t = ((red[0][val][cc] + red[1][val][cc]) + red[2][val][cc]) + red[3][val][cc];
        // val = (n * 2 + i) * 4 + e  -> dW2[n][4 (cc + 32 i) + e];   val = NO * 8 + i * 4 + e -> db1[4 (cc + 32 i) + e]
        const int e = val & 3, i = (val >> 2) & 1, n = val >> 3;
        const int col = 4 * (cc + 32 * i) + e;
        if (col < n_hid) pb[(size_t)n * n_hid + col] = t;      // n == NO is the db1 row
    }
    if (threadIdx.x < NO) {
        const int val = NV + threadIdx.x;
        pb[(size_t)(NO + 1) * n_hid + threadIdx.x] = ((red[0][val][0] + red[1][val][0]) + red[2][val][0]) + red[3][val][0];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Many rows (the sequence model: 19,456), n_hid = 256, n_in = 128: head_thin_bwd_kernel's work AND the first layer's data gradient
// dx = dz1 W1 in one weights-stationary kernel on rowgemm_kernel's schedule (howl_gemm.hip.h): one workgroup per CU keeps W1 in
// registers (wave w: input columns 16 w ..) and streams 16-row tiles; the tile that goes to LDS is not loaded but MADE -- thread
// (wave w, lane) holds y1[row w (+ 8)][4 lane ..], the row's NO output gradients and its four columns of W2, forms
// dz1 = (y1 > 0) * (dy2 W2), writes it to HBM (the first layer's weight gradient reads it) and to the LDS tile, and adds its
// share of dW2 = dy2^T y1, db1 = colsum dz1, db2 = colsum dy2 in registers; the MFMAs then multiply the tile by W1.
// dz1 is never re-read (20 MB), one launch and its ramp disappear (head_thin_bwd 12.0 + rowgemm 19.5 us apart).
// part: head_thin_bwd_kernel's slab layout, one slab per workgroup.
// ---------------------------------------------------------------------------------------------------------
constexpr int HB_THREADS = 512, HB_HID = 256, HB_IN = 128, HB_LDW = HB_HID + 4;
template <int NO>
__global__ __launch_bounds__(HB_THREADS) void head_bwd_rows_kernel(const float* __restrict__ y1, const float* __restrict__ dy2, int rows,
                                                                   const float* __restrict__ w2, const float* __restrict__ w1,
                                                                   float* __restrict__ dz1, float* __restrict__ dx,
                                                                   float* __restrict__ part, int nblocks, HowlCtcMean cm) {
    if ((int)blockIdx.x == nblocks) {      // one extra block: the batch mean of a CTC loss whose launch was left out
        ctc_mean_block(cm.nll, cm.target_lengths, cm.B, cm.loss);
        return;
    }
    constexpr int NV = NO * 4 + 4;          // per-lane partial sums: dW2[n][4 lane + e] (n < NO), db1[4 lane + e]
    __shared__ __attribute__((aligned(16))) float tile[2][16 * HB_LDW];
    __shared__ float red[HB_THREADS / 64][NV][64];
    __shared__ float redd[HB_THREADS / 64][NO];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mi = lane & 15, kq = lane >> 4;
    // A fragments: W1(k = 16 j + 4 kq + e, n = 16 wave + mi) = w1[k * 128 + n]   (reduction index permuted inside a group of 16:
    // see rowgemm_kernel)
    float wv[HB_HID / 16][4];
#pragma unroll
    for (int j = 0; j < HB_HID / 16; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) wv[j][e] = w1[(long)(16 * j + 4 * kq + e) * HB_IN + 16 * wave + mi];
    float4 w2r[NO], aw[NO];
    float ad[NO];
    float4 ab = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int n = 0; n < NO; ++n) {
        w2r[n] = *reinterpret_cast<const float4*>(w2 + (long)n * HB_HID + 4 * lane);
        aw[n] = make_float4(0.f, 0.f, 0.f, 0.f);
        ad[n] = 0.0f;
    }
    const int ntiles = (rows + 15) >> 4;
    struct Pieces {
        float4 y[2];
        float d[2][NO];
    };
    // rows wave and wave + 8 of tile t; unconditional loads from clamped rows
    auto fetch = [&](int t) -> Pieces {
        Pieces p;
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            const long r = min(16 * min(t, ntiles - 1) + wave + 8 * l, rows - 1);
            p.y[l] = *reinterpret_cast<const float4*>(y1 + r * HB_HID + 4 * lane);
#pragma unroll
            for (int n = 0; n < NO; ++n) p.d[l][n] = dy2[r * NO + n];
        }
        return p;
    };
    // makes the dz1 rows of tile t: to LDS buffer `buf`, to HBM, and into the partial sums.  Rows past the end (and a whole tile
    // past the end) were loaded from the last row, so what is stored for them IS the last row, bit for bit -- stored there again
    // (no store under a lane predicate: rowgemm_kernel) -- and they stay out of the sums.
    auto stage = [&](const Pieces& p, int t, int buf) {
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            const int r = 16 * t + wave + 8 * l;
            const float4 v = p.y[l];
            float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int n = 0; n < NO; ++n) {
                z.x = fmaf(p.d[l][n], w2r[n].x, z.x);
                z.y = fmaf(p.d[l][n], w2r[n].y, z.y);
                z.z = fmaf(p.d[l][n], w2r[n].z, z.z);
                z.w = fmaf(p.d[l][n], w2r[n].w, z.w);
            }
            z.x = v.x > 0.0f ? z.x : 0.0f;
            z.y = v.y > 0.0f ? z.y : 0.0f;
            z.z = v.z > 0.0f ? z.z : 0.0f;
            z.w = v.w > 0.0f ? z.w : 0.0f;
            *reinterpret_cast<float4*>(&tile[buf][(wave + 8 * l) * HB_LDW + 4 * lane]) = z;
            const long rr = min(16 * min(t, ntiles - 1) + wave + 8 * l, rows - 1);      // the row fetch() read
            *reinterpret_cast<float4*>(dz1 + rr * HB_HID + 4 * lane) = z;
            if (r < rows) {      // (wave-uniform)
#pragma unroll
                for (int n = 0; n < NO; ++n) {
                    aw[n].x = fmaf(p.d[l][n], v.x, aw[n].x);
                    aw[n].y = fmaf(p.d[l][n], v.y, aw[n].y);
                    aw[n].z = fmaf(p.d[l][n], v.z, aw[n].z);
                    aw[n].w = fmaf(p.d[l][n], v.w, aw[n].w);
                    ad[n] += p.d[l][n];
                }
                ab.x += z.x;
                ab.y += z.y;
                ab.z += z.z;
                ab.w += z.w;
            }
        }
    };
    // dx rows of the tile in buffer `cur`: D[n_local = 4 kq + r][m = mi] -> four consecutive input columns of row 16 t + mi
    auto multiply = [&](int cur) -> float4 {
        f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
        const float* arow = &tile[cur][mi * HB_LDW + 4 * kq];
#pragma unroll
        for (int j = 0; j < HB_HID / 16; j += 2) {
            const float4 a = *reinterpret_cast<const float4*>(arow + 16 * j);
            const float4 b = *reinterpret_cast<const float4*>(arow + 16 * j + 16);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j][0], a.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j + 1][0], b.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j][1], a.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j + 1][1], b.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j][2], a.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j + 1][2], b.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j][3], a.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j + 1][3], b.w, acc1, 0, 0, 0);
        }
        return make_float4(acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]);
    };
    auto store = [&](const float4& o, int t) {
        *reinterpret_cast<float4*>(dx + (long)min(16 * t + mi, rows - 1) * HB_IN + 16 * wave + 4 * kq) = o;
    };
    // schedule: rowgemm_kernel's (two tiles on their way, the next tile made after the MFMAs and before the result stores)
    const int G = nblocks;
    int t = blockIdx.x;
    Pieces p0 = fetch(t);
    Pieces p1 = fetch(t + G);
    stage(p0, t, 0);
    __syncthreads();
    for (; t < ntiles; t += 2 * G) {
        p0 = fetch(t + 2 * G);
        __builtin_amdgcn_sched_barrier(0);      // keep the requests in front of the MFMAs
        const float4 o0 = multiply(0);
        stage(p1, t + G, 1);
        __builtin_amdgcn_sched_barrier(0);
        store(o0, t);
        __syncthreads();
        if (t + G >= ntiles) break;
        p1 = fetch(t + 3 * G);
        __builtin_amdgcn_sched_barrier(0);
        const float4 o1 = multiply(1);
        stage(p0, t + 2 * G, 0);
        __builtin_amdgcn_sched_barrier(0);
        store(o1, t + G);
        __syncthreads();
    }
    // the eight waves' partial sums (rows w, w + 8 of every tile each) meet in LDS, folded in a fixed order
#pragma unroll
    for (int n = 0; n < NO; ++n) {
        red[wave][n * 4 + 0][lane] = aw[n].x;
        red[wave][n * 4 + 1][lane] = aw[n].y;
        red[wave][n * 4 + 2][lane] = aw[n].z;
        red[wave][n * 4 + 3][lane] = aw[n].w;
        if (lane == 0) redd[wave][n] = ad[n];
    }
    red[wave][NO * 4 + 0][lane] = ab.x;
    red[wave][NO * 4 + 1][lane] = ab.y;
    red[wave][NO * 4 + 2][lane] = ab.z;
    red[wave][NO * 4 + 3][lane] = ab.w;
    __syncthreads();
    float* pb = part + (size_t)blockIdx.x * ((size_t)(NO + 1) * HB_HID + NO);
    for (int idx = tid; idx < NV * 64; idx += HB_THREADS) {
        const int val = idx >> 6, ln = idx & 63;
        float tsum = 0.0f;
#pragma unroll
        for (int w_ = 0; w_ < HB_THREADS / 64; ++w_) tsum += red[w_][val][ln];
        // val = 4 n + e -> dW2[n][4 ln + e] (n < NO), db1[4 ln + e] (n == NO)
        pb[(size_t)(val >> 2) * HB_HID + 4 * ln + (val & 3)] = tsum;
    }
    if (tid < NO) {
        float tsum = 0.0f;
#pragma unroll
        for (int w_ = 0; w_ < HB_THREADS / 64; ++w_) tsum += redd[w_][tid];
        pb[(size_t)(NO + 1) * HB_HID + tid] = tsum;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Round 6: head forward + log_softmax / CTC + head backward of the sequence model in ONE launch.  Between
// the two recurrences of a seq-lstm training step sat rowgemm_kernel (y1 = relu(H W1^T + b1), logits) -> ctc_kernel ->
// head_bwd_rows_kernel (dz1, dW2 / db partials, dH = dz1 W1): three launches touching rows that are independent per utterance,
// and two trips of the (rows, 256) hidden activations through HBM (20 MB each way at 512 x 38).  Here a workgroup owns a GROUP
// of U (1 or 2) whole utterances: its U T rows run through the first layer in 16-row tiles on rowgemm_kernel's schedule with
// the results kept in LDS (y1 never leaves the chip), the thin output layer rides as before, then waves 0 .. 2U-1 run the
// utterances' CTC recursions (howl_ctc.hip.h, row pitch 17: <= 8 classes, <= 8 labels; alpha and beta on a wave each) on the logits in
// LDS while the other waves fetch the second set of W1 fragments, then head_bwd_rows_kernel's tile pipeline runs with its input MADE from LDS.
// Every product and sum of a row is taken in the order of the three kernels it replaces: logits, nll, dlogits, dz1 and dH are
// bit-identical to theirs; the per-workgroup slabs of dW2 / db1 / db2 cover different rows (sums in another order).
// Needs the rows of a group in LDS: 16 ceil(U T / 16) x 260 floats, i.e. windows up to ~0.6 s at U = 2 and ~1.2 s at U = 1
// (BASELINE config 4: 0.5 s); longer batches (whole clips) keep the three launches.
// ---------------------------------------------------------------------------------------------------------
constexpr int SH_THREADS = 512, SH_LDY = HB_HID + 4, SH_LDI = HB_IN + 4, SH_RPC = 17, SH_LG = 8;
struct SeqHeadArgs {
    const float* h;            // hidden rows: row (b, t) at h + b * h_souter + t * h_sinner
    long h_souter, h_sinner;
    const float *w1, *b1, *w2, *b2;
    float* y2;                 // (B, T, NO) logits
    const long long* targets;
    long tgt_stride;
    const long long* in_len;
    const long long* tgt_len;
    int blank;
    float* nll;                // (B)
    float* dz1;                // (B T, 256)
    float* dx;                 // (B T, 128)
    float* part;               // [workgroup][(NO + 1) 256 + NO]
    int B, T, U, ngroups;
};
__host__ __device__ inline int seq_head_rt(int U, int T) { return 16 * ((U * T + 15) / 16); }
__host__ __device__ inline size_t seq_head_lds_floats(int U, int T, int n_out) {
    const size_t r0a = (size_t)seq_head_rt(U, T) * SH_LDY, r0b = (size_t)(4 * n_out + 4) * SH_THREADS;
    return (r0a > r0b ? r0a : r0b) + 2 * 16 * SH_LDY + 2 * 8 * 16 * 8 + 2 * (size_t)seq_head_rt(U, T) * SH_LG +
           (size_t)U * (5 * (size_t)T * SH_RPC + 64);
}

template <int NO>
__global__ __launch_bounds__(SH_THREADS) void seq_head_ctc_kernel(SeqHeadArgs a) {
    HIP_DYNAMIC_SHARED(float, lds)
    constexpr int NV = NO * 4 + 4, KG = HB_IN / 16, NT = 2;
    const int T = a.T, U = a.U, RT = seq_head_rt(U, T);
    const size_t r0a = (size_t)RT * SH_LDY, r0b = (size_t)NV * SH_THREADS;
    float* y1s = lds;                                        // [RT][260]; at the very end: the partial sums' meeting place
    float* tiles = lds + (r0a > r0b ? r0a : r0b);            // backward: dz1 tiles [2][16][260]; forward: input tiles [2][16][132]
    float* red2 = tiles + 2 * 16 * SH_LDY;                   // [2][8][16][8]
    float* lg = red2 + 2 * 8 * 16 * 8;                       // [RT][8] logits
    float* dlg = lg + (size_t)RT * SH_LG;                    // [RT][8] d loss / d logits
    float* ctcb = dlg + (size_t)RT * SH_LG;                  // U x (5 T 17 + 64): ctc_pair_*
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mi = lane & 15, kq = lane >> 4;
    const int nbase = wave * 16 * NT;
    // backward partial sums: this thread's rows of every tile of every group of the workgroup
    float4 aw[NO];
    float ad[NO];
    float4 ab = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int n = 0; n < NO; ++n) {
        aw[n] = make_float4(0.f, 0.f, 0.f, 0.f);
        ad[n] = 0.0f;
    }
    for (int i = tid; i < 2 * 16 * SH_LDY; i += SH_THREADS) tiles[i] = 0.0f;
    const int prow = tid >> 5, pcol = 4 * (tid & 31);        // the 16-byte piece of an input tile this thread moves
    for (int grp = blockIdx.x; grp < a.ngroups; grp += gridDim.x) {
        const int R0 = grp * U * T;
        const int nrows = min(U * T, a.B * T - R0);
        const int NTL = (nrows + 15) >> 4;
        const float* w1p = a.w1;
        HOWL_OPAQUE_S(w1p);       // (the fragments of a phase are loaded in that phase: 2 x 64 registers would not fit next to it)
        // ---- forward: y1 = relu(H W1^T + b1) -> LDS, logits = y1 W2^T + b2 -> LDS + HBM (rowgemm_kernel<8, 2, true, NO>) --------
        {
            float wv[NT][KG][4];
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < KG; ++j) {
                    const float4 t4 = *reinterpret_cast<const float4*>(w1p + (long)(nbase + 16 * i + mi) * HB_IN + 16 * j + 4 * kq);
                    wv[i][j][0] = t4.x, wv[i][j][1] = t4.y, wv[i][j][2] = t4.z, wv[i][j][3] = t4.w;
                }
            float4 bv[NT], w2v[NT];
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                bv[i] = *reinterpret_cast<const float4*>(a.b1 + nbase + 16 * i + 4 * kq);
                const float4 t4 = *reinterpret_cast<const float4*>(a.w2 + (long)min(mi, NO - 1) * HB_HID + nbase + 16 * i + 4 * kq);
                w2v[i] = mi < NO ? t4 : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            auto fetch = [&](int k) -> float4 {
                const int r = R0 + min(16 * k + prow, nrows - 1);
                const int b = r / T;
                return *reinterpret_cast<const float4*>(a.h + (long)b * a.h_souter + (long)(r - b * T) * a.h_sinner + pcol);
            };
            auto stage = [&](const float4& v, int buf) { *reinterpret_cast<float4*>(&tiles[buf * (16 * SH_LDI + 4) + prow * SH_LDI + pcol]) = v; };
            float4 p0 = fetch(0);
            __syncthreads();          // the previous group's backward is past the tiles
            stage(p0, 0);
            __syncthreads();
            for (int k = 0; k < NTL; ++k) {
                const int cur = k & 1;
                if (k + 1 < NTL) p0 = fetch(k + 1);
                __builtin_amdgcn_sched_barrier(0);
                f32x4 acc[NT][2];
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    acc[i][0] = {0.0f, 0.0f, 0.0f, 0.0f};
                    acc[i][1] = {0.0f, 0.0f, 0.0f, 0.0f};
                }
                const float* arow = &tiles[cur * (16 * SH_LDI + 4) + mi * SH_LDI + 4 * kq];
#pragma unroll
                for (int j = 0; j < KG; ++j) {
                    const float4 av4 = *reinterpret_cast<const float4*>(arow + 16 * j);
                    const float av[4] = {av4.x, av4.y, av4.z, av4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int i = 0; i < NT; ++i)
                            acc[i][j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i][j][e], av[e], acc[i][j & 1], 0, 0, 0);
                }
                float4 ov[NT];
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    float4 v = make_float4(acc[i][0][0] + acc[i][1][0] + bv[i].x, acc[i][0][1] + acc[i][1][1] + bv[i].y,
                                           acc[i][0][2] + acc[i][1][2] + bv[i].z, acc[i][0][3] + acc[i][1][3] + bv[i].w);
                    ov[i] = make_float4(fmaxf(v.x, 0.0f), fmaxf(v.y, 0.0f), fmaxf(v.z, 0.0f), fmaxf(v.w, 0.0f));
                }
                f32x4 y = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    y = __builtin_amdgcn_mfma_f32_16x16x4f32(w2v[i].x, ov[i].x, y, 0, 0, 0);
                    y = __builtin_amdgcn_mfma_f32_16x16x4f32(w2v[i].y, ov[i].y, y, 0, 0, 0);
                    y = __builtin_amdgcn_mfma_f32_16x16x4f32(w2v[i].z, ov[i].z, y, 0, 0, 0);
                    y = __builtin_amdgcn_mfma_f32_16x16x4f32(w2v[i].w, ov[i].w, y, 0, 0, 0);
                }
                if (kq < 2) *reinterpret_cast<float4*>(&red2[((cur * 8 + wave) * 16 + mi) * 8 + 4 * kq]) = make_float4(y[0], y[1], y[2], y[3]);
                if (k + 1 < NTL) stage(p0, cur ^ 1);
#pragma unroll
                for (int i = 0; i < NT; ++i)
                    *reinterpret_cast<float4*>(&y1s[(size_t)(16 * k + mi) * SH_LDY + nbase + 16 * i + 4 * kq]) = ov[i];
                __syncthreads();      // next tile complete, this tile's partial outputs in red2
                {
                    const int e = tid % (16 * NO), r = e / NO, n = e - r * NO;
                    float yy = a.b2[n];
#pragma unroll
                    for (int w_ = 0; w_ < 8; ++w_) yy += red2[((cur * 8 + w_) * 16 + r) * 8 + n];
                    lg[(16 * k + r) * SH_LG + n] = yy;
                    a.y2[(long)(R0 + min(16 * k + r, nrows - 1)) * NO + n] = yy;
                }
            }
        }
        __syncthreads();
        // ---- log_softmax + CTC of the group's utterances on waves 0 .. 2U-1 (dlogits -> LDS), W1's second set of fragments on all ----
        // two waves per utterance: alpha on the even one, beta on the odd one (howl_ctc.hip.h, ctc_pair_*)
        const int cu = wave >> 1, crole = wave & 1, cb = grp * U + cu;
        const bool cwave = cu < U && cb < a.B;
        float* cbuf = ctcb + (size_t)cu * (5 * (size_t)T * SH_RPC + 64);
        float cnll = 0.0f;
        if (cwave)
            cnll = ctc_pair_recursion<SH_RPC>(crole, lg + (size_t)cu * T * SH_LG, SH_LG, T, NO, a.targets + (size_t)cb * a.tgt_stride,
                                              (int)a.in_len[cb], (int)a.tgt_len[cb], a.blank, a.nll + cb, cbuf, lane);
        HOWL_OPAQUE_S(w1p);
        float wvb[HB_HID / 16][4];
#pragma unroll
        for (int j = 0; j < HB_HID / 16; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) wvb[j][e] = w1p[(long)(16 * j + 4 * kq + e) * HB_IN + 16 * wave + mi];
        float4 w2r[NO];
#pragma unroll
        for (int n = 0; n < NO; ++n) w2r[n] = *reinterpret_cast<const float4*>(a.w2 + (long)n * HB_HID + 4 * lane);
        __syncthreads();      // alpha and beta rows of every utterance of the group in LDS
        if (cwave && crole == 0)
            ctc_pair_grad<SH_RPC>(T, a.B, NO, a.targets + (size_t)cb * a.tgt_stride, (int)a.in_len[cb], (int)a.tgt_len[cb], a.blank, cnll,
                                  dlg + (size_t)cu * T * SH_LG, SH_LG, cbuf, lane);
        __syncthreads();
        // ---- backward (head_bwd_rows_kernel): dz1 = (y1 > 0) (dlogits W2) -> HBM + tile, dH = dz1 W1, partial sums --------------------
        {
            auto stage = [&](int k, int buf) {
#pragma unroll
                for (int l = 0; l < 2; ++l) {
                    const int r = 16 * k + wave + 8 * l;
                    const int rc = min(r, nrows - 1);          // rows past the group's end: the last row again (stored there again)
                    const float4 v = *reinterpret_cast<const float4*>(&y1s[(size_t)rc * SH_LDY + 4 * lane]);
                    float d[NO];
#pragma unroll
                    for (int n = 0; n < NO; ++n) d[n] = dlg[rc * SH_LG + n];
                    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int n = 0; n < NO; ++n) {
                        z.x = fmaf(d[n], w2r[n].x, z.x);
                        z.y = fmaf(d[n], w2r[n].y, z.y);
                        z.z = fmaf(d[n], w2r[n].z, z.z);
                        z.w = fmaf(d[n], w2r[n].w, z.w);
                    }
                    z.x = v.x > 0.0f ? z.x : 0.0f;
                    z.y = v.y > 0.0f ? z.y : 0.0f;
                    z.z = v.z > 0.0f ? z.z : 0.0f;
                    z.w = v.w > 0.0f ? z.w : 0.0f;
                    *reinterpret_cast<float4*>(&tiles[buf * (16 * SH_LDY) + (wave + 8 * l) * SH_LDY + 4 * lane]) = z;
                    *reinterpret_cast<float4*>(a.dz1 + (long)(R0 + rc) * HB_HID + 4 * lane) = z;
                    if (r < nrows) {      // (wave-uniform)
#pragma unroll
                        for (int n = 0; n < NO; ++n) {
                            aw[n].x = fmaf(d[n], v.x, aw[n].x);
                            aw[n].y = fmaf(d[n], v.y, aw[n].y);
                            aw[n].z = fmaf(d[n], v.z, aw[n].z);
                            aw[n].w = fmaf(d[n], v.w, aw[n].w);
                            ad[n] += d[n];
                        }
                        ab.x += z.x;
                        ab.y += z.y;
                        ab.z += z.z;
                        ab.w += z.w;
                    }
                }
            };
            stage(0, 0);
            __syncthreads();
            for (int k = 0; k < NTL; ++k) {
                const int cur = k & 1;
                f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
                const float* arow = &tiles[cur * (16 * SH_LDY) + mi * SH_LDY + 4 * kq];
#pragma unroll
                for (int j = 0; j < HB_HID / 16; j += 2) {
                    const float4 x0 = *reinterpret_cast<const float4*>(arow + 16 * j);
                    const float4 x1 = *reinterpret_cast<const float4*>(arow + 16 * j + 16);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wvb[j][0], x0.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wvb[j + 1][0], x1.x, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wvb[j][1], x0.y, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wvb[j + 1][1], x1.y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wvb[j][2], x0.z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wvb[j + 1][2], x1.z, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wvb[j][3], x0.w, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wvb[j + 1][3], x1.w, acc1, 0, 0, 0);
                }
                const float4 o = make_float4(acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]);
                if (k + 1 < NTL) stage(k + 1, cur ^ 1);
                *reinterpret_cast<float4*>(a.dx + (long)(R0 + min(16 * k + mi, nrows - 1)) * HB_IN + 16 * wave + 4 * kq) = o;
                __syncthreads();
            }
        }
    }
    // the eight waves' partial sums meet in LDS (the y1 region: every wave is past the last barrier), folded in a fixed order
    float* red = lds;
#pragma unroll
    for (int n = 0; n < NO; ++n) {
        red[(wave * NV + n * 4 + 0) * 64 + lane] = aw[n].x;
        red[(wave * NV + n * 4 + 1) * 64 + lane] = aw[n].y;
        red[(wave * NV + n * 4 + 2) * 64 + lane] = aw[n].z;
        red[(wave * NV + n * 4 + 3) * 64 + lane] = aw[n].w;
        if (lane == 0) red2[wave * NO + n] = ad[n];
    }
    red[(wave * NV + NO * 4 + 0) * 64 + lane] = ab.x;
    red[(wave * NV + NO * 4 + 1) * 64 + lane] = ab.y;
    red[(wave * NV + NO * 4 + 2) * 64 + lane] = ab.z;
    red[(wave * NV + NO * 4 + 3) * 64 + lane] = ab.w;
    __syncthreads();
    float* pb = a.part + (size_t)blockIdx.x * ((size_t)(NO + 1) * HB_HID + NO);
    for (int idx = tid; idx < NV * 64; idx += SH_THREADS) {
        const int val = idx >> 6, ln = idx & 63;
        float tsum = 0.0f;
#pragma unroll
        for (int w_ = 0; w_ < SH_THREADS / 64; ++w_) tsum += red[(w_ * NV + val) * 64 + ln];
        pb[(size_t)(val >> 2) * HB_HID + 4 * ln + (val & 3)] = tsum;
    }
    if (tid < NO) {
        float tsum = 0.0f;
#pragma unroll
        for (int w_ = 0; w_ < SH_THREADS / 64; ++w_) tsum += red2[w_ * NO + tid];
        pb[(size_t)(NO + 1) * HB_HID + tid] = tsum;
    }
}

// workspace of one Linear layer's backward on the GEMM path, in floats: split-K slabs of the weight gradient (<= 128) + 256 slabs
// of n_out for the bias column sums
size_t linear_ws_floats(int n_out, int n_in) { return (size_t)HEAD_W1_SPLITS * n_out * (n_in > 1 ? n_in : 1) + (size_t)256 * n_out + 64; }

bool head_is_thin(int n_hid, int n_out) { return n_out >= 1 && n_out <= 8 && n_hid <= HEAD_MAX_HID && (n_hid & 3) == 0; }

// How (and whether) seq_head_ctc_kernel covers a batch: U utterances per group, groups, workgroups (= slabs of partial sums)
struct SeqHeadGeom {
    int U, ngroups, blocks;
    size_t lds_bytes;
};
bool seq_head_geometry(int B, int T, int n_in, int n_hid, int n_out, int max_target, SeqHeadGeom* g) {
    if (n_in != HB_IN || n_hid != HB_HID || n_out < 1 || n_out > 8 || B < 1 || T < 1 || T > CTC_CHUNK) return false;
    if (2 * max_target + 1 > SH_RPC) return false;                       // the CTC rows hold <= 17 states
    if ((long)B * T < rowgemm_min_rows() || (long)B * T >= (1L << 20)) return false;     // few rows: the tile kernels keep more CUs busy
    const char* e = getenv("HOWL_SEQ_HEAD_FUSED");
    if (e != nullptr && e[0] == '0') return false;
    constexpr size_t LDS_MAX = 160 * 1024;
    int U = 0;
    for (int u = 2; u >= 1; --u)
        if (seq_head_lds_floats(u, T, n_out) * sizeof(float) <= LDS_MAX) {
            U = u;
            break;
        }
    if (U == 0) return false;
    g->U = U;
    g->ngroups = (B + U - 1) / U;
    g->blocks = std::min(std::min(g->ngroups, howl_num_cus()), HEAD_BWD_BLOCKS);
    g->lds_bytes = seq_head_lds_floats(U, T, n_out) * sizeof(float);
    return true;
}

// Which recurrence pair runs: four sequences per workgroup (v_mfma 4x4x1_16b: 93 / 73 us per 38-step launch, one workgroup
// per CU) while that leaves at most two rounds of workgroups, sixteen per workgroup (16x16x4: 213 / 198 us) beyond --
// B <= 8 x CUs = 2048 on this part.  HOWL_LSTM_ROWS=4|16 forces one (tests exercise both).
bool lstm_rows16(int B, int T) {
    if ((size_t)B * (size_t)(T + 1) * G4 * sizeof(float) >= ((size_t)1 << 32)) return true;   // the 4-row kernels index bytes in 32 bits
    const char* env = getenv("HOWL_LSTM_ROWS");
    if (env != nullptr && env[0] == '1') return true;
    if (env != nullptr && env[0] == '4') return false;
    return B > 8 * howl_num_cus();
}

}  // namespace

extern "C" {

size_t howl_lstm_workspace_bytes(int B, int T) {
    // packed W_hh of the 16-row recurrences (2 x 64K floats) + bias sum (512) + split-K scratch of the W_hh gradient
    // (128 x 512 x 128) + the W_ih gradient's own scratch (128 x 512 x 96 at most) + the bias column sums' (256 x 512):
    // three regions, so that the three final slab sums can run as one launch (the bias region holds one slab per workgroup
    // of the four-sequence recurrence, or the 256 of the column-sum kernel)
    const size_t bias_slabs = (size_t)(B + 3) / 4 > 256 ? (size_t)(B + 3) / 4 : 256;
    return ((size_t)2 * 16 * 64 * 64 + G4 + (size_t)LSTM_WGRAD_SPLITS * G4 * HID + (size_t)LSTM_WGRAD_SPLITS * G4 * LSTM_MAX_IN +
            bias_slabs * G4) * sizeof(float) + 1024;
}

// (the four-sequence recurrence multiplies x_t W_ih^T itself when M = 40: no projection launch, gx unused)
static bool lstm_fuse_x(const HowlLstmParams* p, int B, int T, int M, int xf) {
    return !lstm_rows16(B, T) && M == 40 && (reinterpret_cast<uintptr_t>(p->w_ih) & 15) == 0 &&
           (size_t)B * (size_t)xf * M * sizeof(float) < ((size_t)1 << 32) && getenv("HOWL_LSTM_NO_FUSED_X") == nullptr;
}

size_t howl_lstm_needs_gx(const HowlLstmParams* p, int B, int T, int M, int x_frames) {
    if (p == nullptr || p->w_ih == nullptr) return 1;
    return lstm_fuse_x(p, B, T, M, x_frames > 0 ? x_frames : T) ? 0 : 1;
}

static int lstm_fwd_impl(const HowlLstmParams* p, const float* x, int B, int T, int M, const long long* lengths, const float* h0,
                         const float* c0, const HowlLstmSaved* sv, float* hT, float* cT, void* ws, size_t ws_bytes,
                         const HowlLogmelArgs* next, hipStream_t stream) {
    HOWL_REQUIRE(p && x && sv && hT && cT && ws, "howl_lstm_fwd: null pointer");
    LogmelLaunch ll{};
    if (next != nullptr) {
        const int rc = logmel_prepare(next->pcm, next->B, next->L, next->ld, next->fbp, next->M, next->log_eps, next->zmuv, next->out,
                                      next->layout, &ll);
        if (rc != HOWL_OK) return rc;
    }
    HOWL_REQUIRE(B >= 1 && T >= 1 && M >= 1, "howl_lstm_fwd: bad shape");
    HOWL_REQUIRE(sv->t_out >= 1 && sv->t_out <= T, "howl_lstm_fwd: t_out=%d outside 1..T", sv->t_out);
    if (ws_bytes < howl_lstm_workspace_bytes(B, T)) {
        howl_set_error("howl_lstm_fwd: workspace too small");
        return HOWL_E_WORKSPACE;
    }
    float* pf = static_cast<float*>(ws);
    float* pb = pf + 16 * 64 * 64;
    float* bsum = pb + 16 * 64 * 64;
    const bool rows16 = lstm_rows16(B, T);
    // the 16-row recurrences take packed W_hh fragments and a folded bias (one launch); the 4-row ones read the parameters
    // in place and add the bias themselves: gx = x W_ih^T (+ b_ih + b_hh)   (B*T, 512), K = M
    if (rows16)
        hipLaunchKernelGGL(lstm_pack_kernel, dim3(16 * 64 * 64 / 256), dim3(256), 0, stream, p->w_hh, pf, pb, p->b_ih, p->b_hh, bsum);
    const int xf = sv->x_frames > 0 ? sv->x_frames : T;
    HOWL_REQUIRE(xf >= T, "howl_lstm_fwd: x_frames=%d < T=%d", xf, T);
    const bool fuse_x = lstm_fuse_x(p, B, T, M, xf);
    HOWL_REQUIRE(fuse_x || sv->gx != nullptr, "howl_lstm_fwd: saved->gx is NULL but this shape runs the projection GEMM "
                                              "(howl_lstm_needs_gx)");
    if (!fuse_x)
        gemm(stream, true, x, xf == T ? lin(M) : RowMap{T, (long)xf * M, M}, 1, lin(0), p->w_ih, lin(1), M, B * T, G4, M, 1,
             rows16 ? bsum : nullptr, 0, sv->gx, G4, 0);
    if (sv->t_out < T)   // rows of steps that never run are read (times zero) by the weight-gradient GEMM: keep them finite
        hipMemsetAsync(sv->hseq, 0, (size_t)B * (T + 1) * HID * sizeof(float), stream);
    // h_{t-1} W_hh^T of every step (+ x_t W_ih^T where the recurrence multiplies it itself)
    // the next batch's frontend rides in this launch when the recurrence leaves half of the device idle (HOWL_LSTM_RIDE_LOGMEL=0:
    // as its own launch behind the recurrence, as it is whenever the four-sequence kernel with the fused projection does not run)
    const bool ride = next != nullptr && fuse_x && (B + 3) / 4 <= howl_num_cus() / 2 && next->M <= 4 * NG_BANDED &&
                      (getenv("HOWL_LSTM_RIDE_LOGMEL") == nullptr || getenv("HOWL_LSTM_RIDE_LOGMEL")[0] != '0');
    {
    HowlProfScope prof("lstm_fwd", stream, 2.0 * (HID + (fuse_x ? M : 0)) * G4 * (double)B * sv->t_out);
    if (fuse_x) {
        const int nrec = (B + 3) / 4;
        if (ride) {
            const int nb = std::max(1, std::min(ll.n_quads, howl_num_cus() - nrec));
            hipLaunchKernelGGL((lstm_fwd4_kernel<40, true>), dim3(nrec + nb), dim3(F8_THREADS), 0, stream, x, p->w_ih, xf, p->w_hh, p->b_ih,
                               p->b_hh, lengths, h0, c0, sv->gates, sv->c, sv->hseq, hT, cT, B, T, sv->t_out, nrec, ll, nb);
        } else {
            hipLaunchKernelGGL(lstm_fwd4_kernel<40>, dim3(nrec), dim3(F8_THREADS), 0, stream, x, p->w_ih, xf, p->w_hh, p->b_ih, p->b_hh,
                               lengths, h0, c0, sv->gates, sv->c, sv->hseq, hT, cT, B, T, sv->t_out, nrec, LogmelLaunch{}, 0);
        }
    } else if (!rows16) {
        hipLaunchKernelGGL(lstm_fwd4_kernel<0>, dim3((B + 3) / 4), dim3(F8_THREADS), 0, stream, (const float*)sv->gx,
                           (const float*)nullptr, 0, p->w_hh, p->b_ih, p->b_hh, lengths, h0, c0, sv->gates, sv->c, sv->hseq, hT, cT, B,
                           T, sv->t_out, (B + 3) / 4, LogmelLaunch{}, 0);
    } else {
        const size_t lds_fwd = (size_t)(2 * 16 * HS + 2 * 16 * 6 * HID) * sizeof(float);
        hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fwd);
        hipLaunchKernelGGL(lstm_fwd_kernel, dim3((B + 15) / 16), dim3(LSTM_THREADS), lds_fwd, stream, (const float*)sv->gx,
                           (const float*)pf, lengths, h0, c0, sv->gates, sv->c, sv->hseq, hT, cT, B, T, sv->t_out);
    }
    }
    if (next != nullptr && !ride) {
        const int rc = howl_logmel_fwd(next->pcm, next->B, next->L, next->ld, next->fbp, next->M, next->log_eps, next->zmuv, next->out,
                                       next->layout, stream);
        if (rc != HOWL_OK) return rc;
    }
    HOWL_CHECK_LAUNCH("howl_lstm_fwd");
    return HOWL_OK;
}

int howl_lstm_fwd(const HowlLstmParams* p, const float* x, int B, int T, int M, const long long* lengths, const float* h0,
                  const float* c0, const HowlLstmSaved* sv, float* hT, float* cT, void* ws, size_t ws_bytes,
                  hipStream_t stream) {
    return lstm_fwd_impl(p, x, B, T, M, lengths, h0, c0, sv, hT, cT, ws, ws_bytes, nullptr, stream);
}

int howl_lstm_fwd_next(const HowlLstmParams* p, const float* x, int B, int T, int M, const long long* lengths, const float* h0,
                       const float* c0, const HowlLstmSaved* sv, float* hT, float* cT, void* ws, size_t ws_bytes,
                       const HowlLogmelArgs* next, hipStream_t stream) {
    HOWL_REQUIRE(next != nullptr, "howl_lstm_fwd_next: null HowlLogmelArgs (howl_lstm_fwd is the call without one)");
    return lstm_fwd_impl(p, x, B, T, M, lengths, h0, c0, sv, hT, cT, ws, ws_bytes, next, stream);
}

// body of howl_lstm_bwd; with `jobs` the wide weight gradients are collected instead of launched (howl_seq_lstm_bwd runs them
// together with the head's), and the slab folds go to `sums` (flushed by the caller)
static int lstm_bwd_impl(const HowlLstmParams* p, const float* x, int B, int T, int M, const long long* lengths, const float* c0,
                         const HowlLstmSaved* sv, const float* dy, const float* dhT, const float* dcT, const HowlLstmGrads* g,
                         void* ws, size_t ws_bytes, hipStream_t stream, SlabSums& sums, WgradJobs* jobs,
                         const WgradJob* ride = nullptr) {
    HOWL_REQUIRE(p && x && sv && g && ws, "howl_lstm_bwd: null pointer");
    HOWL_REQUIRE(dy || dhT, "howl_lstm_bwd: no incoming gradient");
    if (ws_bytes < howl_lstm_workspace_bytes(B, T)) {
        howl_set_error("howl_lstm_bwd: workspace too small");
        return HOWL_E_WORKSPACE;
    }
    float* pf = static_cast<float*>(ws);
    float* pb = pf + 16 * 64 * 64;
    float* scratch = pb + 16 * 64 * 64 + G4;
    float* scratch_ih = scratch + (size_t)LSTM_WGRAD_SPLITS * G4 * HID;
    float* scratch_b = scratch_ih + (size_t)LSTM_WGRAD_SPLITS * G4 * LSTM_MAX_IN;
    const int Tout = sv->t_out;
    const bool rows16 = lstm_rows16(B, T);
    {
    // dG_t W_hh of every step (+ the job that rides along)
    HowlProfScope prof("lstm_bwd", stream, 2.0 * HID * G4 * (double)B * Tout + (ride != nullptr ? 2.0 * (double)ride->M * ride->N * ride->K : 0.0));
    if (!rows16) {
        const int nrec = (B + 3) / 4;
        hipLaunchKernelGGL(lstm_bwd4_kernel, dim3(nrec + (ride != nullptr ? wgrad_job_blocks(*ride) : 0)), dim3(B8_THREADS), 0, stream, dy,
                           dhT, dcT, p->w_hh, lengths, (const float*)sv->gates, (const float*)sv->c, c0, sv->dgates, scratch_b, B, T,
                           Tout, nrec, ride != nullptr ? *ride : WgradJob{});
    }
    else
        hipLaunchKernelGGL(lstm_bwd_kernel, dim3((B + 15) / 16), dim3(LSTM_THREADS), 0, stream, dy, dhT, dcT, (const float*)pb,
                           lengths, (const float*)sv->gates, (const float*)sv->c, c0, sv->dgates, B, T, Tout);
    }
    // dW_ih = dG^T X, dW_hh = dG^T H_prev (hseq rows t = 0..t_out-1 of each utterance), db = column sums of dG.  Steps
    // t >= t_out never ran and their dG rows are never written: the reductions walk rows (b, t < t_out) only.
    const bool full = Tout == T;
    const RowMap rows_g = full ? lin(G4) : RowMap{Tout, (long)T * G4, G4};
    const int xf = sv->x_frames > 0 ? sv->x_frames : T;
    HOWL_REQUIRE(xf >= T, "howl_lstm_bwd: x_frames=%d < T=%d", xf, T);
    const RowMap rows_x = (full && xf == T) ? lin(M) : RowMap{Tout, (long)xf * M, M};
    const int rows = B * Tout;
    // 256 rows per K slice (up to 128 slices): the 512-row slices of the generic rule leave ~1 block per CU on these shapes
    HOWL_REQUIRE(M <= LSTM_MAX_IN, "howl_lstm_bwd: M=%d input features exceed the workspace layout (max %d)", M, LSTM_MAX_IN);
    // one job for both products where the shape allows (dG -- 40 MB at 512 x 38 -- read and staged once): howl_gemm.hip.h
    if (!wgrad_dual_gemm(sv->dgates, rows_g, G4, sv->hseq, RowMap{Tout, (long)(T + 1) * HID, HID}, x, rows_x, M, rows, scratch, g->w_hh,
                         scratch_ih, g->w_ih, LSTM_WGRAD_SPLITS, &sums, jobs)) {
        wgrad_gemm(stream, sv->dgates, rows_g, G4, x, rows_x, M, rows, scratch_ih, g->w_ih, LSTM_WGRAD_SPLITS, 256, &sums, jobs);
        wgrad_gemm(stream, sv->dgates, rows_g, G4, sv->hseq, RowMap{Tout, (long)(T + 1) * HID, HID}, HID, rows, scratch, g->w_hh,
                   LSTM_WGRAD_SPLITS, 256, &sums, jobs);
    }
    if (!rows16)   // the four-sequence recurrence left one slab of step-and-sequence sums per workgroup
        sums.add(scratch_b, (B + 3) / 4, G4, g->b_ih, g->b_hh);
    else
        colsum(stream, sv->dgates, rows_g, rows, G4, scratch_b, g->b_ih, g->b_hh, 256, 64, &sums);
    return HOWL_OK;
}


int howl_lstm_bwd(const HowlLstmParams* p, const float* x, int B, int T, int M, const long long* lengths, const float* c0,
                  const HowlLstmSaved* sv, const float* dy, const float* dhT, const float* dcT, const HowlLstmGrads* g,
                  void* ws, size_t ws_bytes, hipStream_t stream) {
    SlabSums sums;
    const int rc = lstm_bwd_impl(p, x, B, T, M, lengths, c0, sv, dy, dhT, dcT, g, ws, ws_bytes, stream, sums, nullptr);
    if (rc != HOWL_OK) return rc;
    if (!sums.flush(stream)) return HOWL_E_ARG;
    HOWL_CHECK_LAUNCH("howl_lstm_bwd");
    return HOWL_OK;
}

size_t howl_head_workspace_bytes(int n_in, int n_hid, int n_out) {
    // [first layer: up to 128 split-K slabs of n_hid x n_in] [second layer, thin: HEAD_BWD_BLOCKS slabs of (n_out + 1) n_hid + n_out]
    // (other shapes: the GEMM path's slabs of both layers)
    const size_t first = (size_t)HEAD_W1_SPLITS * n_hid * n_in;
    const size_t thin = (size_t)HEAD_BWD_BLOCKS * ((size_t)(n_out + 1) * n_hid + n_out);
    const size_t general = linear_ws_floats(n_hid, n_in) + linear_ws_floats(n_out, n_hid);
    const size_t a = first + thin + 64, b = general + 64;
    return (a > b ? a : b) * sizeof(float);
}

int howl_head_fwd(const HowlHeadParams* p, const float* x, int rows_inner, long s_outer, long s_inner, int rows, int n_in,
                  int n_hid, int n_out, float* y1, float* y2, hipStream_t stream) {
    HOWL_REQUIRE(p && p->w1 && p->b1 && p->w2 && p->b2 && x && y1 && y2, "howl_head_fwd: null pointer");
    HOWL_REQUIRE(rows >= 1 && n_in >= 1 && n_hid >= 1 && n_out >= 1 && rows_inner >= 1, "howl_head_fwd: bad shape");
    const bool is_thin = head_is_thin(n_hid, n_out);
    const RowGemmThin second{p->w2, p->b2, y2};
    bool second_done = false;     // many rows: the second layer rides in the first layer's launch (rowgemm_kernel)
    gemm(stream, true, x, RowMap{rows_inner, s_outer, s_inner}, 1, lin(0), p->w1, lin(1), n_in, rows, n_hid, n_in, 1, p->b1, 1,
         y1, n_hid, 0, is_thin ? &second : nullptr, n_out, &second_done);
    if (second_done) {
    } else if (is_thin) {
        int blocks = (rows + 15) / 16;
        const int cap = 4 * howl_num_cus();
        blocks = blocks > cap ? cap : blocks;
#define HOWL_HEAD_OUT(NO) \
    case NO: hipLaunchKernelGGL(head_out_kernel<NO>, dim3(blocks), dim3(256), 0, stream, (const float*)y1, rows, n_hid, p->w2, p->b2, y2); break;
        switch (n_out) {
            HOWL_HEAD_OUT(1) HOWL_HEAD_OUT(2) HOWL_HEAD_OUT(3) HOWL_HEAD_OUT(4) HOWL_HEAD_OUT(5) HOWL_HEAD_OUT(6) HOWL_HEAD_OUT(7)
            HOWL_HEAD_OUT(8)
        }
#undef HOWL_HEAD_OUT
    } else {
        gemm(stream, true, y1, lin(n_hid), 1, lin(0), p->w2, lin(1), n_hid, rows, n_out, n_hid, 1, p->b2, 0, y2, n_out, 0);
    }
    HOWL_CHECK_LAUNCH("howl_head_fwd");
    return HOWL_OK;
}

static int head_bwd_impl(const HowlHeadParams* p, const float* x, int rows_inner, long s_outer, long s_inner, int rows, int n_in,
                         int n_hid, int n_out, const float* y1, const float* dy2, float* dz1, float* dx, const HowlHeadGrads* g,
                         const HowlCtcMean* ctc_mean, void* ws, size_t ws_bytes, hipStream_t stream, SlabSums& sums, WgradJobs* jobs) {
    HOWL_REQUIRE(ctc_mean == nullptr || (ctc_mean->nll && ctc_mean->target_lengths && ctc_mean->loss && ctc_mean->B >= 1),
                 "howl_head_bwd: incomplete HowlCtcMean");
    const HowlCtcMean cm = ctc_mean != nullptr ? *ctc_mean : HowlCtcMean{nullptr, nullptr, 0, nullptr};
    HOWL_REQUIRE(p && p->w1 && p->w2 && x && ((y1 && dy2) || (!y1 && !dy2)) && dz1 && g && g->w1 && g->b1 && g->w2 && g->b2 && ws,
                 "howl_head_bwd: null pointer");
    HOWL_REQUIRE(rows >= 1 && n_in >= 1 && n_hid >= 1 && n_out >= 1 && rows_inner >= 1, "howl_head_bwd: bad shape");
    if (ws_bytes < howl_head_workspace_bytes(n_in, n_hid, n_out)) {
        howl_set_error("howl_head_bwd: workspace too small");
        return HOWL_E_WORKSPACE;
    }
    float* first = static_cast<float*>(ws);
    const RowMap xm{rows_inner, s_outer, s_inner};
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    bool dx_done = false;
    if (dy2 == nullptr) {
        // howl_seq_head_ctc ran the rows already (dz1, dx and one slab of partial sums per workgroup are in place): fold its slabs
        SeqHeadGeom sg;
        HOWL_REQUIRE(y1 == nullptr && rows % rows_inner == 0 && seq_head_geometry(rows / rows_inner, rows_inner, n_in, n_hid, n_out, 0, &sg),
                     "howl_head_bwd: dy2 == NULL means the rows were run by howl_seq_head_ctc, which does not cover this shape");
        float* thin = first + (size_t)HEAD_W1_SPLITS * n_hid * n_in;
        if (ctc_mean != nullptr) sums.mean = SlabMean{cm.nll, cm.target_lengths, cm.B, cm.loss};     // rides in the call's fold launch
        const long slab = (long)(n_out + 1) * n_hid + n_out;
        sums.add_strided(thin, sg.blocks, slab, (long)n_out * n_hid, g->w2);
        sums.add_strided(thin + (size_t)n_out * n_hid, sg.blocks, slab, n_hid, g->b1);
        sums.add_strided(thin + (size_t)(n_out + 1) * n_hid, sg.blocks, slab, n_out, g->b2);
        dx_done = true;
    } else if (head_is_thin(n_hid, n_out) && dx != nullptr && n_hid == HB_HID && n_in == HB_IN && rows >= rowgemm_min_rows() &&
        rows < (1 << 20) && al16(y1) && al16(dz1) && al16(dx) && al16(p->w2) && getenv("HOWL_GEMM_NO_ROWGEMM") == nullptr) {
        // many rows: second layer's backward + ReLU mask + dx = dz1 W1 in one launch, one slab of partial sums per workgroup
        float* thin = first + (size_t)HEAD_W1_SPLITS * n_hid * n_in;
        const int ntiles = (rows + 15) / 16;
        const int blocks = std::min(std::min(ntiles, howl_num_cus()), HEAD_BWD_BLOCKS);
        HowlProfScope prof("gemm", stream, 2.0 * (double)rows * n_in * n_hid);
#define HOWL_HEAD_BWD_ROWS(NO) \
    case NO: hipLaunchKernelGGL(head_bwd_rows_kernel<NO>, dim3(blocks + (ctc_mean != nullptr ? 1 : 0)), dim3(HB_THREADS), 0, stream, y1, dy2, rows, p->w2, p->w1, dz1, dx, thin, blocks, cm); break;
        switch (n_out) {
            HOWL_HEAD_BWD_ROWS(1) HOWL_HEAD_BWD_ROWS(2) HOWL_HEAD_BWD_ROWS(3) HOWL_HEAD_BWD_ROWS(4) HOWL_HEAD_BWD_ROWS(5)
            HOWL_HEAD_BWD_ROWS(6) HOWL_HEAD_BWD_ROWS(7) HOWL_HEAD_BWD_ROWS(8)
        }
#undef HOWL_HEAD_BWD_ROWS
        const long slab = (long)(n_out + 1) * n_hid + n_out;
        sums.add_strided(thin, blocks, slab, (long)n_out * n_hid, g->w2);
        sums.add_strided(thin + (size_t)n_out * n_hid, blocks, slab, n_hid, g->b1);
        sums.add_strided(thin + (size_t)(n_out + 1) * n_hid, blocks, slab, n_out, g->b2);
        dx_done = true;
    } else if (head_is_thin(n_hid, n_out)) {
        float* thin = first + (size_t)HEAD_W1_SPLITS * n_hid * n_in;
        int rpb = (rows + HEAD_BWD_BLOCKS - 1) / HEAD_BWD_BLOCKS;
        rpb = (rpb + 7) / 8 * 8;
        const int blocks = (rows + rpb - 1) / rpb;
#define HOWL_HEAD_BWD(NO) \
    case NO: hipLaunchKernelGGL(head_thin_bwd_kernel<NO>, dim3(blocks + (ctc_mean != nullptr ? 1 : 0)), dim3(256), 0, stream, y1, dy2, rows, n_hid, rpb, p->w2, dz1, thin, blocks, cm); break;
        switch (n_out) {
            HOWL_HEAD_BWD(1) HOWL_HEAD_BWD(2) HOWL_HEAD_BWD(3) HOWL_HEAD_BWD(4) HOWL_HEAD_BWD(5) HOWL_HEAD_BWD(6) HOWL_HEAD_BWD(7)
            HOWL_HEAD_BWD(8)
        }
#undef HOWL_HEAD_BWD
        const long slab = (long)(n_out + 1) * n_hid + n_out;
        sums.add_strided(thin, blocks, slab, (long)n_out * n_hid, g->w2);
        sums.add_strided(thin + (size_t)n_out * n_hid, blocks, slab, n_hid, g->b1);
        sums.add_strided(thin + (size_t)(n_out + 1) * n_hid, blocks, slab, n_out, g->b2);
    } else {
        // general shapes: second layer by the GEMM path (its own folds), ReLU mask, then the first layer below
        if (ctc_mean != nullptr) hipLaunchKernelGGL(ctc_mean_only_kernel, dim3(1), dim3(256), 0, stream, cm);
        float* ws2 = first + linear_ws_floats(n_hid, n_in);
        float* scratch_b2 = ws2 + (size_t)HEAD_W1_SPLITS * n_out * n_hid;
        gemm(stream, true, dy2, lin(n_out), 1, lin(0), p->w2, lin(n_hid), 1, rows, n_hid, n_out, 1, nullptr, 0, dz1, n_hid, 0);
        hipLaunchKernelGGL(relu_bwd_kernel, dim3(1024), dim3(256), 0, stream, (const float*)dz1, y1, (long)rows * n_hid, dz1);
        wgrad_gemm(stream, dy2, lin(n_out), n_out, y1, lin(n_hid), n_hid, rows, ws2, g->w2, 64, 512, &sums);
        colsum(stream, dy2, lin(n_out), rows, n_out, scratch_b2, g->b2, nullptr, 256, 64, &sums);
        float* scratch_b1 = first + (size_t)HEAD_W1_SPLITS * n_hid * n_in;
        colsum(stream, dz1, lin(n_hid), rows, n_hid, scratch_b1, g->b1, nullptr, 256, 64, &sums);
    }
    if (dx != nullptr && !dx_done)   // dx = dz1 W1
        gemm(stream, true, dz1, lin(n_hid), 1, lin(0), p->w1, lin(n_in), 1, rows, n_in, n_hid, 1, nullptr, 0, dx, n_in, 0);
    wgrad_gemm(stream, dz1, lin(n_hid), n_hid, x, xm, n_in, rows, first, g->w1, HEAD_W1_SPLITS, 512, &sums, jobs);
    return HOWL_OK;
}

int howl_head_bwd(const HowlHeadParams* p, const float* x, int rows_inner, long s_outer, long s_inner, int rows, int n_in,
                  int n_hid, int n_out, const float* y1, const float* dy2, float* dz1, float* dx, const HowlHeadGrads* g,
                  const HowlCtcMean* ctc_mean, void* ws, size_t ws_bytes, hipStream_t stream) {
    SlabSums sums;
    const int rc = head_bwd_impl(p, x, rows_inner, s_outer, s_inner, rows, n_in, n_hid, n_out, y1, dy2, dz1, dx, g, ctc_mean, ws,
                                 ws_bytes, stream, sums, nullptr);
    if (rc != HOWL_OK) return rc;
    if (!sums.flush(stream)) return HOWL_E_ARG;
    HOWL_CHECK_LAUNCH("howl_head_bwd");
    return HOWL_OK;
}

int howl_seq_head_ctc_supported(int B, int T, int n_in, int n_hid, int n_out, int max_target_length) {
    SeqHeadGeom g;
    return seq_head_geometry(B, T, n_in, n_hid, n_out, max_target_length, &g) ? 1 : 0;
}

int howl_seq_head_ctc(const HowlHeadParams* p, const float* x, long s_outer, long s_inner, int B, int T, int n_in, int n_hid, int n_out,
                      const long long* targets, long tgt_stride, int max_target_length, const long long* input_lengths,
                      const long long* target_lengths, int blank, float* y2, float* nll, float* dz1, float* dhs, void* head_ws,
                      size_t head_ws_bytes, hipStream_t stream) {
    HOWL_REQUIRE(p && p->w1 && p->b1 && p->w2 && p->b2 && x && targets && input_lengths && target_lengths && y2 && nll && dz1 && dhs &&
                     head_ws, "howl_seq_head_ctc: null pointer");
    HOWL_REQUIRE(blank >= 0 && blank < n_out && max_target_length >= 0, "howl_seq_head_ctc: bad blank / target length");
    SeqHeadGeom g;
    HOWL_REQUIRE(seq_head_geometry(B, T, n_in, n_hid, n_out, max_target_length, &g),
                 "howl_seq_head_ctc: B=%d T=%d (%d -> %d -> %d, targets <= %d) is outside the fused launch's range "
                 "(howl_seq_head_ctc_supported; use howl_head_fwd + howl_ctc_loss + howl_seq_lstm_bwd)", B, T, n_in, n_hid, n_out,
                 max_target_length);
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    HOWL_REQUIRE(al16(x) && al16(p->w1) && al16(p->b1) && al16(p->w2) && al16(dz1) && al16(dhs) && (s_outer & 3) == 0 && (s_inner & 3) == 0,
                 "howl_seq_head_ctc: 16-byte aligned operands");
    if (head_ws_bytes < howl_head_workspace_bytes(n_in, n_hid, n_out)) {
        howl_set_error("howl_seq_head_ctc: workspace too small");
        return HOWL_E_WORKSPACE;
    }
    float* thin = static_cast<float*>(head_ws) + (size_t)HEAD_W1_SPLITS * n_hid * n_in;
    const SeqHeadArgs a{x, s_outer, s_inner, p->w1, p->b1, p->w2, p->b2, y2, targets, tgt_stride, input_lengths, target_lengths, blank,
                        nll, dz1, dhs, thin, B, T, g.U, g.ngroups};
    HowlProfScope prof("gemm", stream, 4.0 * (double)B * T * n_in * n_hid);
#define HOWL_SEQ_HEAD(NO)                                                                                                        \
    case NO: {                                                                                                                   \
        static thread_local size_t granted[16] = {};                                                                             \
        howl_raise_lds(reinterpret_cast<const void*>(seq_head_ctc_kernel<NO>), g.lds_bytes, granted, "howl_seq_head_ctc");        \
        hipLaunchKernelGGL(seq_head_ctc_kernel<NO>, dim3(g.blocks), dim3(SH_THREADS), g.lds_bytes, stream, a);                   \
    } break;
    switch (n_out) {
        HOWL_SEQ_HEAD(1) HOWL_SEQ_HEAD(2) HOWL_SEQ_HEAD(3) HOWL_SEQ_HEAD(4) HOWL_SEQ_HEAD(5) HOWL_SEQ_HEAD(6) HOWL_SEQ_HEAD(7)
        HOWL_SEQ_HEAD(8)
    }
#undef HOWL_SEQ_HEAD
    HOWL_CHECK_LAUNCH("howl_seq_head_ctc");
    return HOWL_OK;
}

int howl_seq_lstm_bwd(const HowlHeadParams* hp, int n_hid, int n_out, const float* y1, const float* dy2, float* dz1, float* dhs,
                      const HowlHeadGrads* hg, const HowlCtcMean* ctc_mean, void* head_ws, size_t head_ws_bytes,
                      const HowlLstmParams* p, const float* x, int B, int T, int M, const long long* lengths, const float* c0,
                      const HowlLstmSaved* sv, const HowlLstmGrads* g, void* ws, size_t ws_bytes, const HowlAdamW* adamw,
                      hipStream_t stream) {
    HOWL_REQUIRE(sv && dhs, "howl_seq_lstm_bwd: null pointer");
    HOWL_REQUIRE(adamw == nullptr || (adamw->p && adamw->g && adamw->m && adamw->v && adamw->n >= 1 && adamw->step >= 1),
                 "howl_seq_lstm_bwd: incomplete HowlAdamW");
    HOWL_REQUIRE(sv->t_out == T, "howl_seq_lstm_bwd: t_out=%d != T=%d (use howl_head_bwd + howl_lstm_bwd)", sv->t_out, T);
    SlabSums sums;
    WgradJobs jobs;
    // the head reads the hidden states h_1 .. h_T in place: row (b, t) of hseq (B, T + 1, 128) at offset 128
    int rc = head_bwd_impl(hp, sv->hseq + HID, T, (long)(T + 1) * HID, HID, B * T, HID, n_hid, n_out, y1, dy2, dz1, dhs, hg, ctc_mean,
                           head_ws, head_ws_bytes, stream, sums, &jobs);
    if (rc != HOWL_OK) return rc;
    // The head's first-layer weight gradient (dz1^T H) depends on nothing the LSTM's backward produces, and the four-sequence
    // recurrence occupies (B + 3) / 4 CUs for its 38-81 dependent steps: while that leaves at least half of the device idle the
    // job's blocks ride in the recurrence's launch (lstm_bwd4_kernel: 18 us of the 65-us job launch at 512 x 38).  HOWL_LSTM_RIDE:
    // "lane" = the same job on the library's side lane instead (two cross-queue waits on the critical path: measured 9 us worse),
    // "0" = behind the recurrence with the other weight gradients (round 4).
    const char* ride_env = getenv("HOWL_LSTM_RIDE");
    const bool idle_half = jobs.count == 1 && !lstm_rows16(B, T) && (B + 3) / 4 <= howl_num_cus() / 2;
    const bool as_ride = idle_half && jobs.j[0].tn == 128 && (ride_env == nullptr || ride_env[0] == '1');
    HowlSideLane* lane = idle_half && ride_env != nullptr && ride_env[0] == 'l' ? howl_side_lane() : nullptr;
    WgradJob ride{};
    if (as_ride) {
        ride = jobs.j[0];
        jobs.count = 0;
        jobs.flops = 0.0;
    } else if (lane != nullptr) {
        howl_lane_fork(lane, stream);
        wgrad_jobs_flush(lane->stream, jobs);
    }
    rc = lstm_bwd_impl(p, x, B, T, M, lengths, c0, sv, dhs, nullptr, nullptr, g, ws, ws_bytes, stream, sums, &jobs, as_ride ? &ride : nullptr);
    if (rc != HOWL_OK) {
        if (lane != nullptr) howl_lane_join(lane, stream);
        return rc;
    }
    wgrad_jobs_flush(stream, jobs);
    if (lane != nullptr) howl_lane_join(lane, stream);
    if (adamw != nullptr && sums.covers(adamw->g, adamw->n) && getenv("HOWL_NO_FOLD_ADAMW") == nullptr) {
        // every gradient of the model leaves this call through the fold: the optimiser step rides in it
        const double bc1 = 1.0 - pow((double)adamw->beta1, (double)adamw->step), bc2 = 1.0 - pow((double)adamw->beta2, (double)adamw->step);
        const SlabAdamW opt{adamw->p, adamw->g, adamw->m, adamw->v,
                            HowlAdamWCoef{adamw->lr, adamw->beta1, adamw->beta2, adamw->eps, adamw->weight_decay, (float)bc1,
                                          (float)sqrt(bc2), adamw->grad_scale}, 1};
        if (!sums.flush(stream, &opt)) return HOWL_E_ARG;
    } else {
        if (!sums.flush(stream)) return HOWL_E_ARG;
        if (adamw != nullptr) {
            const int rc2 = howl_adamw_step(adamw->p, adamw->g, adamw->m, adamw->v, adamw->n, adamw->lr, adamw->beta1, adamw->beta2,
                                            adamw->eps, adamw->weight_decay, adamw->step, adamw->grad_scale, stream);
            if (rc2 != HOWL_OK) return rc2;
        }
    }
    HOWL_CHECK_LAUNCH("howl_seq_lstm_bwd");
    return HOWL_OK;
}


}  // extern "C"
