// res8 classifier forward + backward for gfx950 (MI355X).
//
// Replaces the ATen op chain of howl/model/cnn.py:113-145 (Res8: conv0 -> ReLU -> AvgPool(3,4) ->
// 6 x [conv3x3 45->45, ReLU, (+residual on even i), BatchNorm2d(affine=False)] -> mean -> Linear) and its
// autograd backward (convolution_backward, threshold_backward, native_batch_norm_backward,
// avg_pool2d_backward), as driven by training/run/pretrain_gsc.py:126-133 and training/run/train.py:288-302.
//
// Layouts (HBM, fp32): activations s_i are (B, 45, H, 10) exactly like the reference's NCHW tensors
// (H = T/3, P = H*10 positions); weights keep the reference's (45,45,3,3) / (45,1,3,3) / (C,45) shapes so
// state_dicts are interchangeable (howl/workspace.py:31-67).  Maps that do not fit this tile -- 80 mel bins (20 pooled columns:
// settings.py:32's default), more than 83 frames (27 pooled rows) -- run as STRIPS: blocks of (45, <= 27, 10) that fetch their
// neighbours' edge columns / rows on the way into LDS (HaloSlot, StripGeom below; separate template instances, the plain
// kernels are compiled exactly as before).
//
// conv3x3 45->45 (12 of the 13 GFLOP-heavy launches per step: 6 forward, 6 dgrad) is an implicit GEMM on
// v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains, 157 TFLOP/s peak):
//     D[position 16][cout 16] += A[position][k] * B[k][cout],   k = (cin block of 4, tap)
//   * one workgroup = 12 wavefronts = 3 cout tiles x 4 position groups; the packed weight fragments (102 k-steps of 4:
//     11 blocks of 4 input channels x 9 taps + channel 44's nine taps as a 3-step tail) are staged ONCE per workgroup into
//     LDS (83 KB; round 1 kept them in VGPRs, which spilled address registers and serialised the staging code);
//   * one utterance's whole (48, H+1, 12) zero-haloed input map lives in LDS next to them (64.7 KB at H=27, single
//     buffered: weights + one tile fill the 160 KB); the next utterance's loads are issued in four bursts between segments
//     of the K loop and the previous layer's BatchNorm is applied on the way into LDS ((s - mean) * rstd), so normalised
//     activations are never materialised in HBM; in training the launch also folds the producer's BatchNorm partials in its
//     prologue (no finalize launch between two convolutions);
//   * epilogue fuses ReLU, the residual add, the store, and the per-channel sum / sum-of-squares that the next
//     BatchNorm needs (per-workgroup partials, folded deterministically by the consumer).
// wgrad is the transposed GEMM (M = cout, N = tap x cin, K = positions) with the 45x405 accumulators resident
// in registers across all utterances of a workgroup, written once as per-workgroup partials.
#include "howl_common.hip.h"
#include "../../include/howl_hip.h"

namespace {

struct HowlPtrs6 {
    float* p[6];
};
constexpr int MAX_WINDOWS = 64;   // howl_res8_fwd_long: 64 windows of 13 new pooled rows each = clips up to ~2,500 frames (25 s)
struct HowlWinRows {
    int lo[MAX_WINDOWS], hi[MAX_WINDOWS];
};
struct HowlBnBuffers {
    float* running_mean;
    float* running_var;
    long long* num_batches;
};
// Statistics partials of the PRODUCING convolution handed to the consuming one: every workgroup of the consumer folds the
// [nparts][2][48] rows itself (fp64, fixed order: identical in all workgroups) while its weights are in flight, instead of
// a separate one-block kernel between the two launches; workgroup 0 also publishes mean / rstd for the later readers
// (backward pass, head) and updates the running buffers.
struct BnFold {
    const float* part;   // nullptr: statistics come ready-made (in_stats), nothing to fold
    int nparts;
    double count;
    float* stats_out;    // [2][48]
    HowlBnBuffers bn;
};


constexpr int NMAP = 45;         // res8 feature maps (cnn.py:110)
constexpr int CP = 48;           // channels padded to 3 MFMA tiles
constexpr int PW = 10;           // pooled width = 40 mels / 4
constexpr int WP = 12;           // LDS row pitch: 10 + left/right halo
constexpr int CONV_THREADS = 768;
constexpr int KFULL = 11;        // full input-channel blocks (4 channels x 9 taps = 9 k-steps each): channels 0..43
constexpr int KSTEPS = 9 * KFULL + 3;   // + channel 44 alone: its 9 taps as 3 k-steps (taps 4s + k); 102 instead of the 108
                                 // of a zero-padded 12th block: 5.6 % fewer MFMAs in the forward / dgrad K loops
constexpr int MAX_H = 27;
constexpr int MAX_ROW_STRIPS = 1024;   // row strips per utterance (howl_res8_fwd / _bwd): 82,944 frames; nothing in the kernels depends on the count
constexpr float BN_EPS = 1e-5f;
constexpr float BN_MOMENTUM = 0.1f;

__host__ __device__ inline int chan_stride(int H, bool own_rows = false) {
    // (H+1) rows of 12 (top halo + data; the bottom halo is the next channel's top halo), padded so that
    // CS = 17 (mod 32): position-major reads (forward/dgrad) and channel-major reads (wgrad) both spread over banks
    // (own_rows: H+2 rows -- row strips of a long map fetch REAL halo rows from their neighbours, see StripGeom)
    int cs = (H + (own_rows ? 2 : 1)) * WP;
    int pad = (17 - (cs % 32) + 32) % 32;
    return cs + pad;
}
__host__ __device__ inline int tile_floats(int H, bool own_rows = false) { return CP * chan_stride(H, own_rows) + 32; }
constexpr int WPW = 13;          // row pitch of the weight-gradient kernel's x tile (see wgrad_body)
// weight-gradient GEMM (wgrad_body): M = cout, N = (tap, cin) FLATTENED: column idx = 45 * tap + cin, 405 columns in 26 tiles of 16 (rounds 1-3 padded every tap to 48
// columns: 27 tiles).  78 accumulator chains (26 N tiles x 3 cout tiles) instead of 81: with two whole N tiles per wave and the
// six chains of tiles 24 and 25 dealt one each to waves 0, 4, 1, 5, 2, 3, the SIMDs carry 20 / 20 / 19 / 19 chains (was 21 / 21 /
// 21 / 18): the role is matrix-pipe bound, so that is 4.8 % of its time.
constexpr int WNCOL = 16 * 26;          // partial row: [48 cout][WNCOL]
constexpr int WTAPS = 9 * NMAP;         // 405 real columns
__host__ __device__ inline int wgrad_rounds(int H) { return (H + 3) / 4; }

// ---------------------------------------------------------------------------------------------------------
// weight packing: (45,45,3,3) -> per-wave MFMA B fragments  wp[nt][kstep][lane]
//   forward : B[k][n] = w[cout = 16nt + n][cin = 4*c0 + k][tap]
//   dgrad   : B[k][n] = w[cout = 4*c0 + k][cin = 16nt + n][8 - tap]   (transposed, spatially flipped)
// ---------------------------------------------------------------------------------------------------------
constexpr int PACK_ELEMS = 3 * KSTEPS * 64;   // fragments of one layer, one direction
__device__ __forceinline__ void pack_weights_one(const HowlPtrs6& w, float* __restrict__ wp_fwd, float* __restrict__ wp_bwd,
                                                 int layer, int mode, int idx) {
    const int lane = idx & 63;
    const int ks = (idx >> 6) % KSTEPS;
    const int nt = idx / (64 * KSTEPS);
    int kk, tap;
    if (ks < 9 * KFULL) {                 // block c0 of four channels, one tap per k-step
        const int c0 = ks / 9;
        tap = ks - 9 * c0;
        kk = 4 * c0 + (lane >> 4);
    } else {                              // channel 44: k = lane >> 4 walks four taps per k-step (taps 9..11 are padding)
        tap = 4 * (ks - 9 * KFULL) + (lane >> 4);
        kk = 4 * KFULL;
    }
    const int n = 16 * nt + (lane & 15);
    float v = 0.0f;
    if (kk < NMAP && n < NMAP && tap < 9) {
        const float* wl = w.p[layer];
        v = (mode == 0) ? wl[(n * NMAP + kk) * 9 + tap] : wl[(kk * NMAP + n) * 9 + (8 - tap)];
    }
    float* dst = (mode == 0 ? wp_fwd : wp_bwd) + (size_t)layer * (3 * KSTEPS * 64);
    dst[idx] = v;
}

#define HOWL_PROBE(cfg_, wave_, lane_, slot_) ((void)0)

// ---------------------------------------------------------------------------------------------------------
// shared pieces of the MFMA kernels
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void zero_lds(float* p, int n, int tid, int nthreads) {   // p 16-byte aligned
    const int n4 = n >> 2;
    for (int i = tid; i < n4; i += nthreads) reinterpret_cast<float4*>(p)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int i = 4 * n4 + tid; i < n; i += nthreads) p[i] = 0.0f;
}

// BatchNorm partial statistics travel as a TRANSPOSED matrix part[column][nps] (column < 2*48: sums, then second moments;
// row = producing workgroup, nps = row count rounded up to 4), so that a consumer wave folds eight columns with coalesced
// 16-byte loads: lane = (column group c8 = lane >> 3, slice = lane & 7) reads rows 4*(slice + 8j)..+3 of its column for
// j = 0, 1, ... (eight full 128-byte lines per wave instruction), adds in fp64 and meets its 7 neighbours through three
// shuffles.  Every lane of a column group returns the column total; the order is fixed, so every workgroup (and every
// run) gets the same bits.
__host__ __device__ inline int part_stride(int nparts) { return (nparts + 3) & ~3; }
__device__ __forceinline__ double fold_part_column(const float* __restrict__ part, int nps, int nparts, int column, int lane) {
    const int slice = lane & 7;
    const float* src = part + (size_t)column * nps;
    double acc = 0.0;
    for (int r0 = 4 * slice; r0 < nparts; r0 += 8 * 32) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = r0 + 32 * j;
            v[j] = *reinterpret_cast<const float4*>(src + (r < nps ? r : 0));      // clamped: all eight in flight
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = r0 + 32 * j;
            acc += (r + 0 < nparts) ? (double)v[j].x : 0.0;
            acc += (r + 1 < nparts) ? (double)v[j].y : 0.0;
            acc += (r + 2 < nparts) ? (double)v[j].z : 0.0;
            acc += (r + 3 < nparts) ? (double)v[j].w : 0.0;
        }
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    acc += __shfl_xor(acc, 4);
    return acc;
}

// One wave's share of an utterance: NTW position tiles (j = t0, t0 + ts, ...: t0 = position group + 4 * slice, ts = 4 * slices
// when `slices` workgroups share an utterance's position tiles, see conv3x3_body) of cout tile nt, all NTW accumulator
// chains advanced together so that one weight fragment read feeds NTW MFMAs and the chains hide each other's latency.
//   A[position][k] from the zero-haloed activation tile, B[k][cout] from the packed weights in LDS.
template <int NTW>
struct KCursor {
    const lds_f32* ap[NTW];
    const lds_f32* bp;
    float a[NTW], b;      // operands of the NEXT k-step, requested one step ahead (k_prime / k_run)
};

// The K loop is software-pipelined by hand: the operands of k-step s+1 are requested before the MFMAs of step s (as the weight
// gradient's K loop has always done), and they travel in the cursor across the staging bursts between two segments.  Left to the
// compiler, the third tap of every row was a ds_read_b32 issued right in front of the MFMA that needs it.
template <int NTW>
__device__ __forceinline__ void k_prime(KCursor<NTW>& k) {
#pragma unroll
    for (int i = 0; i < NTW; ++i) k.a[i] = k.ap[i][0];
    k.b = k.bp[0];
}

template <int NTW, int TS>
__device__ __forceinline__ void k_begin(KCursor<NTW>& k, f32x4 (&acc)[NTW], const lds_f32* tile, const lds_f32* wl,
                                        int CS, int P, int t0, int lane) {
    constexpr int ts = TS;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        int m = 16 * (t0 + ts * i) + (lane & 15);
        m = m < P ? m : P - 1;  // the last tile may overhang: clamp the read, the store is masked
        const int h = m / PW;
        k.ap[i] = tile + (lane >> 4) * CS + h * WP + (m - h * PW);
        HOWL_OPAQUE_LDS(k.ap[i]);
        acc[i] = {0.0f, 0.0f, 0.0f, 0.0f};
    }
    k.bp = wl + lane;
    HOWL_OPAQUE_LDS(k.bp);
}

// `groups` channel groups (4 input channels x 9 taps each) of the K loop
template <int NTW>
__device__ __forceinline__ void k_run(KCursor<NTW>& k, f32x4 (&acc)[NTW], int CS, int groups) {
    // two K groups per trip: the segments of conv_loop are 2, 2, 2 | 2, 2, 1 groups long, i.e. straight-line code; the operand
    // reads of a group are then scheduled under the MFMAs of the one before it (one group per trip: +16 us per c3 step)
#pragma unroll 2
    for (int g = 0; g < groups; ++g) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            float na[NTW], nb;
            if (tap < 8) {
                const int off = ((tap + 1) / 3) * WP + ((tap + 1) % 3);
#pragma unroll
                for (int i = 0; i < NTW; ++i) na[i] = k.ap[i][off];
                nb = k.bp[(tap + 1) * 64];
            } else {            // the first tap of the next group (past the last group of a phase: discarded, see k_prime)
                k.bp += 9 * 64;
#pragma unroll
                for (int i = 0; i < NTW; ++i) k.ap[i] += 4 * CS;
#pragma unroll
                for (int i = 0; i < NTW; ++i) na[i] = k.ap[i][0];
                nb = k.bp[0];
            }
#pragma unroll
            for (int i = 0; i < NTW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(k.a[i], k.b, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NTW; ++i) k.a[i] = na[i];
            k.b = nb;
        }
    }
}

// The last input channel (44) after the KFULL full blocks: three k-steps whose four k lanes groups take four different TAPS
// of that one channel, so the A operand's tap offset depends on the lane group: `dl[s]` = (offset of tap 4s + g) - g * CS
// relative to the cursor (which carries the lane group's channel offset g * CS of the full blocks).
template <int NTW>
__device__ __forceinline__ void k_tail(KCursor<NTW>& k, f32x4 (&acc)[NTW], const int (&dl)[3]) {
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
        const float b = k.bp[s3 * 64];
#pragma unroll
        for (int i = 0; i < NTW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(k.ap[i][dl[s3]], b, acc[i], 0, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Tile staging UNDER the K loop (round 4).  Rounds 1-3 staged an utterance's whole map between two barriers (registers ->
// LDS with the K loop stopped: ~3 k cycles per utterance with one tensor, and what made fusing the BatchNorm / ReLU backward
// into the data gradient's loads a loss: three tensors per element, all of it exposed).  The K loop walks the input channels
// in order, so the tile is reused in halves instead:
//   phase A = K groups 0..5  (channels 0..23)          -- meanwhile channels 24..44 of THIS utterance go to LDS
//   barrier
//   phase B = K groups 6..10 + channel 44's tail       -- meanwhile channels 0..23 of the NEXT utterance go to LDS
//   barrier, epilogue
// (same two barriers per utterance as before; a region is only overwritten after the barrier that ends its last reader).  The
// global loads of a slot are issued two K groups (~8 k cycles) ahead of its LDS write, three slots at most in flight per thread.
// What a tile is made of is described by StageCfg:
//   forward           x = (|s_{i-1}| - mean) * rstd                                     (one tensor)
//   data gradient     dz_i as stored by bn_relu_bwd_kernel                              (one tensor; HOWL_RES8_BWD_FUSED=0)
//   data gradient,    dz_i = mask_i * ds_i,  ds_i = rstd * (dx_i - m1 - xhat * m2) + dskip   (three tensors: dx_i, s_i, dskip)
//   fused             i.e. native_batch_norm_backward + the skip gradient + threshold_backward applied on the way into LDS:
//                     dz_i never exists in HBM and the elementwise pass between two layers of the backward pass is gone;
//                     ds_i (the skip gradient two layers down, and conv0's) is written from here on the layers that have one.
// ---------------------------------------------------------------------------------------------------------
constexpr int SPLIT_C = 24;                 // channels [0, 24) = K groups 0..5, [24, 45) = groups 6..10 + the tail
constexpr int KG_A = SPLIT_C / 4;           // K groups of phase A
constexpr int KG_B = KFULL - KG_A;          // full K groups of phase B
constexpr int NS0 = 5, NS1 = 4;             // slots per thread: 24 P / 2 <= 3240 and 21 P / 2 <= 2835 float2 over 768 threads
static_assert(CONV_THREADS * NS0 >= SPLIT_C * MAX_H * PW / 2 && CONV_THREADS * NS1 >= (NMAP - SPLIT_C) * MAX_H * PW / 2, "slots");
static_assert(KG_A == 6 && KG_B == 5, "the segment schedule of conv_loop is written for 6 + 5 K groups");

// Falling wave priority along a phase (3, 2, 1 over its three K segments): a wave that is ahead yields the matrix pipe to the
// ones behind it, so that the waves of a SIMD reach the barrier together instead of the oldest one finishing at 2/3 of the
// phase and the last one running alone, at half the pipe rate (tools/probe_step4.py; -1 us per forward launch).
#define HOWL_STAIR(p_) __builtin_amdgcn_s_setprio(p_)

struct StageCfg {
    const float* a;       // forward: s_{i-1}; data gradient: dz_i (plain) or dx_i (fused; nullptr: broadcast of dpool / P, layer 6)
    const float* s;       // fused: s_i (sign bit / sign = ReLU mask, |s| -> xhat)
    const float* k;       // fused: ds_{i+2} (skip gradient into s_i) or nullptr
    float* ds;            // fused: ds_i out, or nullptr
    const float* dpool;   // fused, a == nullptr: (B, 48) pooled gradient
    float invP;
    bool fused;
    bool even;            // fused: layer i has a residual add (mask in the sign bit of s_i) or not (mask = s_i > 0)
    bool affine;          // forward: normalise on load
};

struct SlotVal {
    float2 a, s, k;
};

// slot j of a region (channels c0 .. c0 + nch - 1) moves float2 number tid + 768 j of the region to its place in the zero-haloed
// tile; destination and channel depend only on the thread: bits 0..19 LDS float offset, 20..25 channel, -1 = no such element
template <int NSL>
__device__ __forceinline__ void region_slots(int (&pk)[NSL], int c0, int nch, int P, int CS, int tid) {
    // integer divisions by run-time values cost ~25 instructions each, eighteen of them 2.5 k cycles of every workgroup's prologue:
    // floor(e / P) as (int)((e + 0.5) * (1 / P)) -- exact: the quotient's fraction is a multiple of 1 / P, so the half step keeps
    // the product >= 0.5 / P = 1.8e-3 away from an integer, float rounding of a value <= 48 is 3e-6
    const float invP = 1.0f / (float)P;
#pragma unroll
    for (int j = 0; j < NSL; ++j) {
        const int e = 2 * (tid + j * CONV_THREADS);
        const int c = (int)(((float)e + 0.5f) * invP);
        const int p = e - c * P;
        const int h = (int)(((float)p + 0.5f) * 0.1f);
        const int w = p - h * PW;
        pk[j] = (e < nch * P) ? (((c0 + c) * CS + (h + 1) * WP + (w + 1)) | ((c0 + c) << 20)) : -1;
    }
}
// the same, plus bit j of `padbits` = slot j lies in a row >= hv (StripGeom: rows beyond a last row strip's valid ones)
template <int NSL>
__device__ __forceinline__ void region_slots_pad(int (&pk)[NSL], int c0, int nch, int P, int CS, int tid, int hv, unsigned& padbits) {
    region_slots<NSL>(pk, c0, nch, P, CS, tid);
#pragma unroll
    for (int j = 0; j < NSL; ++j) {
        const int e = 2 * (tid + j * CONV_THREADS);
        const int p = e % P;
        if (pk[j] >= 0 && p / PW >= hv) padbits |= 1u << j;
    }
}

// `base` = float offset of (utterance, region) in every tensor of the configuration (uniform: the addresses are "scalar base
// + 32-bit lane offset", no 64-bit address lives in a VGPR).  Unconditional loads from clamped offsets (a load under a lane
// predicate waits for the one before it): slots without an element re-read the region's first.
template <int MODE, int HALO = 0>
__device__ __forceinline__ void slot_load(SlotVal& v, const StageCfg& cfg, size_t base, int i2, int pkj, int b) {
    HOWL_OPAQUE_V(pkj);
    const unsigned off = pkj >= 0 ? 8u * (unsigned)i2 : 0u;
    if (MODE == 1 && cfg.fused) {
        if (cfg.a != nullptr) {
            v.a = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(cfg.a + base) + off);
        } else {
            const unsigned c4 = pkj >= 0 ? 4u * (unsigned)(pkj >> 20) : 0u;
            const float g = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(cfg.dpool + (size_t)(HALO == 1 ? b >> 1 : b) * CP) + c4) * cfg.invP;
            v.a = make_float2(g, g);
        }
        v.s = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(cfg.s + base) + off);
        v.k = cfg.k != nullptr ? *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(cfg.k + base) + off)
                               : make_float2(0.0f, 0.0f);
    } else {
        v.a = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(cfg.a + base) + off);
    }
}

// BatchNorm backward + skip gradient + ReLU mask of one element pair.  bn_relu_bwd_kernel's expression
//     ds = rstd * (g - m1 - xhat * m2) + k,   xhat = (|s| - mean) * rstd
// regrouped around per-channel constants (bwd_fold_to_lds): ds = A g + (Bc |s| + Cc) + k -- two fused multiply-adds and an add
// per element instead of seven operations (staging instructions displace matrix instructions on this part).
// lm = [A | Bc | Cc][48] in LDS.
__device__ __forceinline__ void bn_relu_bwd_pair(const SlotVal& v, const float* lmc, bool even, float2& ds, float2& dz) {
    const float A = lmc[0], Bc = lmc[CP], Cc = lmc[2 * CP];
    ds.x = fmaf(A, v.a.x, fmaf(Bc, fabsf(v.s.x), Cc)) + v.k.x;
    ds.y = fmaf(A, v.a.y, fmaf(Bc, fabsf(v.s.y), Cc)) + v.k.y;
    // layers with a residual add keep their ReLU mask in the sign bit of s (conv_epilogue), the others in its sign
    const float t0 = even ? -v.s.x : v.s.x, t1 = even ? -v.s.y : v.s.y;
    dz.x = t0 > 0.0f ? ds.x : 0.0f;
    dz.y = t1 > 0.0f ? ds.y : 0.0f;
}

template <int MODE>
__device__ __forceinline__ void slot_write(const SlotVal& v, const StageCfg& cfg, size_t base, int i2, int pkj, float* tile,
                                           const float* lm, bool zero = false /* a row beyond the strip's valid ones: see StripGeom */) {
    HOWL_OPAQUE_V(pkj);
    if (pkj < 0) return;
    const int c = pkj >> 20;
    float v0 = v.a.x, v1 = v.a.y;
    if (MODE == 0) {
        // stored activations are non-negative (sums of ReLU outputs); layers with a residual add keep the ReLU mask of
        // their own convolution in the sign bit (conv_epilogue), hence the fabs
        v0 = fabsf(v0);
        v1 = fabsf(v1);
        if (cfg.affine) {   // xhat = (|s| - mean) * rstd as one fused multiply-add: lm = [-mean * rstd | rstd]
            const float sh = lm[c], r = lm[CP + c];
            v0 = fmaf(v0, r, sh);
            v1 = fmaf(v1, r, sh);
        }
    } else if (cfg.fused) {
        float2 ds, dz;
        bn_relu_bwd_pair(v, lm + c, cfg.even, ds, dz);
        if (cfg.ds != nullptr)
            *reinterpret_cast<float2*>(reinterpret_cast<char*>(cfg.ds + base) + 8u * (unsigned)i2) = ds;
        v0 = dz.x;
        v1 = dz.y;
    }
    if (zero) v0 = v1 = 0.0f;
    float* d = tile + (pkj & 0xFFFFF);
    d[0] = v0;
    d[1] = v1;
}

// ---------------------------------------------------------------------------------------------------------
// Wide maps (HALO; NUM_MELS = 80, the reference's stock default, settings.py:32): 20 pooled columns do not fit the tile, so an
// utterance is kept as TWO strips of 10 columns, each a (45, H, 10) block of its own -- "virtual utterance" v = 2 b + strip,
// the layout every kernel here already walks -- and the one thing a strip lacks is its neighbour's edge column, which takes
// the place of the zero halo on that side: column 0 of strip 1 into tile column 11 of strip 0, column 9 of strip 0 into tile
// column 0 of strip 1.  That is one more staging slot per thread and region (<= 24 channels x 27 rows = 648 elements <= 768
// threads), read with a stride of 10 floats from the neighbour's block and pushed through the same arithmetic as the rest of
// the tile.  A workgroup's utterances v, v + nblk, ... keep their parity (the launchers make nblk even), so the other side's
// halo column stays at the zeros of the prologue.  Everything else -- K loop, epilogue, statistics, partials -- is unchanged.
// ---------------------------------------------------------------------------------------------------------
struct HaloSlot {
    int pk;   // LDS float offset of tile column 0 of the element's row (bits 0..19) | channel << 20; -1 = no element
    int g;    // float offset of column 0 of that row in an utterance's (45, P) map
};
__device__ __forceinline__ HaloSlot halo_slot(int c0, int nch, int H, int P, int CS, int tid) {
    const int c = tid / H, h = tid - c * H;
    HaloSlot hs;
    hs.pk = (tid < nch * H) ? (((c0 + c) * CS + (h + 1) * WP) | ((c0 + c) << 20)) : -1;
    hs.g = (tid < nch * H) ? (c0 + c) * P + h * PW : 0;
    return hs;
}
// `nbase` = float offset of the NEIGHBOUR strip's block (uniform), gcol = its edge column (9: left neighbour, 0: right one)
template <int MODE>
__device__ __forceinline__ void halo_load(SlotVal& v, const StageCfg& cfg, size_t nbase, const HaloSlot& hs, int gcol, int b) {
    const unsigned off = hs.pk >= 0 ? 4u * (unsigned)(hs.g + gcol) : 0u;
    v.s = v.k = make_float2(0.0f, 0.0f);
    if (MODE == 1 && cfg.fused) {
        if (cfg.a != nullptr) {
            v.a.x = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(cfg.a + nbase) + off);
        } else {
            const unsigned c4 = hs.pk >= 0 ? 4u * (unsigned)(hs.pk >> 20) : 0u;
            v.a.x = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(cfg.dpool + (size_t)(b >> 1) * CP) + c4) * cfg.invP;
        }
        v.s.x = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(cfg.s + nbase) + off);
        if (cfg.k != nullptr) v.k.x = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(cfg.k + nbase) + off);
    } else {
        v.a.x = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(cfg.a + nbase) + off);
    }
    v.a.y = v.a.x;
}
// lcol = tile column of the halo (0: left, WP - 1: right); the neighbour writes its own ds, so nothing goes back to HBM here
template <int MODE>
__device__ __forceinline__ void halo_write(const SlotVal& v, const StageCfg& cfg, const HaloSlot& hs, int lcol, float* tile,
                                           const float* lm) {
    if (hs.pk < 0) return;
    const int c = hs.pk >> 20;
    float v0 = v.a.x;
    if (MODE == 0) {
        v0 = fabsf(v0);
        if (cfg.affine) v0 = fmaf(v0, lm[CP + c], lm[c]);
    } else if (cfg.fused) {
        float2 ds, dz;
        bn_relu_bwd_pair(v, lm + c, cfg.even, ds, dz);
        v0 = dz.x;
    }
    tile[(hs.pk & 0xFFFFF) + lcol] = v0;
}

// ---------------------------------------------------------------------------------------------------------
// Long maps (HALO == 2; training windows beyond 83 frames, cnn.py:127-145 accepts any T): more than 27 pooled rows do not
// fit the tile either, so an utterance is cut into nr ROW strips of H rows (and, at 80 mel bins, ns = 2 column strips each):
// virtual utterance v = (b * nr + r) * ns + c, every one a (45, H, 10) block.  Row strips fetch REAL halo rows -- the last row
// of strip r - 1 above, the first of strip r + 1 below, twelve columns each: the two corners come from the diagonal
// neighbours -- into a tile that has a top AND a bottom halo row per channel (chan_stride(H, true)); strips without a
// neighbour on a side get zeros written there (a workgroup walks strips of every kind).  H * nr may exceed the map: the LAST row
// strip then owns hv_last < H rows, and its rows beyond are "outside the image": zero on the way into every tile (they hold
// whatever the previous layer's launch computed there), left out of every sum over positions (statistics, pooled sums,
// weight gradients), never read by a neighbour.  The column halo of HALO == 1 rides along when ns == 2.
// ---------------------------------------------------------------------------------------------------------
struct StripGeom {
    int nr, ns, hv_last;
};
struct StripPos {
    int r, c, hv, bu;   // row strip, column strip, valid rows, utterance
};
__device__ __forceinline__ StripPos strip_pos(const StripGeom& g, int v, int H) {
    StripPos p;
    const int q = v / g.ns;
    p.c = v - q * g.ns;
    p.bu = q / g.nr;
    p.r = q - p.bu * g.nr;
    p.hv = (p.r == g.nr - 1) ? g.hv_last : H;
    return p;
}
struct EdgeSlot {
    int pk;   // LDS float offset of the element in the tile (top halo row for row slots) | channel << 20; -1 = no element
    int g;    // float offset inside the SOURCE block: channel * P + (column halo: h * 10; row halo: source column)
    int aux;  // column halo: the element's row h; row halo: column-strip delta of the source block (-1, 0, +1)
};
__device__ __forceinline__ EdgeSlot col_edge_slot(int c0, int nch, int H, int P, int CS, int tid) {
    const int c = tid / H, h = tid - c * H;
    const bool ok = tid < nch * H;
    return EdgeSlot{ok ? (((c0 + c) * CS + (h + 1) * WP) | ((c0 + c) << 20)) : -1, ok ? (c0 + c) * P + h * PW : 0, ok ? h : 0};
}
__device__ __forceinline__ EdgeSlot row_edge_slot(int c0, int nch, int P, int CS, int tid) {
    const int c = tid / WP, w = tid - c * WP;              // tile column w = map column w - 1
    const bool ok = tid < nch * WP;
    const int dc = w == 0 ? -1 : (w == WP - 1 ? 1 : 0);
    const int col = w == 0 ? PW - 1 : (w == WP - 1 ? 0 : w - 1);
    return EdgeSlot{ok ? (((c0 + c) * CS + w) | ((c0 + c) << 20)) : -1, ok ? (c0 + c) * P + col : 0, ok ? dc : 0};
}
// one element of block `blk` at float offset `off` inside it, through the configuration's staging arithmetic; !valid: zero
template <int MODE>
__device__ __forceinline__ void edge_load(SlotVal& v, const StageCfg& cfg, int blk, int off, bool valid, int chan, int bu, int P) {
    const size_t idx = valid ? (size_t)blk * NMAP * P + (size_t)off : 0;
    v.s = v.k = make_float2(0.0f, 0.0f);
    if (MODE == 1 && cfg.fused) {
        v.a.x = cfg.a != nullptr ? cfg.a[idx] : cfg.dpool[(size_t)bu * CP + chan] * cfg.invP;
        v.s.x = cfg.s[idx];
        if (cfg.k != nullptr) v.k.x = cfg.k[idx];
    } else {
        v.a.x = cfg.a[idx];
    }
    v.a.y = v.a.x;
}
template <int MODE>
__device__ __forceinline__ void edge_write(const SlotVal& v, const StageCfg& cfg, int pk, int lds_extra, bool valid, float* tile,
                                           const float* lm) {
    if (pk < 0) return;
    const int c = pk >> 20;
    float v0 = v.a.x;
    if (MODE == 0) {
        v0 = fabsf(v0);
        if (cfg.affine) v0 = fmaf(v0, lm[CP + c], lm[c]);
    } else if (cfg.fused) {
        float2 ds, dz;
        bn_relu_bwd_pair(v, lm + c, cfg.even, ds, dz);
        v0 = dz.x;
    }
    tile[(pk & 0xFFFFF) + lds_extra] = valid ? v0 : 0.0f;
}
// what a HALO == 2 workgroup carries through its utterance loop: geometry, the edge slots of the two channel regions, the
// pad bits of the regular slots (rows >= hv_last: zeroed in the last row strips)
struct GridCtx {
    StripGeom sg;
    EdgeSlot col[2], row[2];
    unsigned pad[2];
};
// all three edges of region `reg` for strip v: loads (three SlotVals), then writes
template <int MODE>
__device__ __forceinline__ void grid_load(SlotVal (&e)[3], const StageCfg& cfg, const GridCtx& gx, int reg, int v, int H, int P) {
    const StripPos p = strip_pos(gx.sg, v, H);
    {   // column halo (ns == 2): the neighbour strip of the same row strip, its edge column; rows >= hv are outside the image
        const EdgeSlot& es = gx.col[reg];
        const bool valid = gx.sg.ns == 2 && es.pk >= 0 && es.aux < p.hv;
        edge_load<MODE>(e[0], cfg, v ^ 1, es.g + (p.c ? PW - 1 : 0), valid, es.pk >= 0 ? es.pk >> 20 : 0, p.bu, P);
    }
    const EdgeSlot& rs = gx.row[reg];
    const int cc = p.c + rs.aux;
    const bool colok = rs.pk >= 0 && cc >= 0 && cc < gx.sg.ns;
    const int chan = rs.pk >= 0 ? rs.pk >> 20 : 0;
    edge_load<MODE>(e[1], cfg, v - gx.sg.ns + rs.aux, rs.g + (H - 1) * PW, colok && p.r > 0, chan, p.bu, P);            // row above
    edge_load<MODE>(e[2], cfg, v + gx.sg.ns + rs.aux, rs.g, colok && p.r < gx.sg.nr - 1, chan, p.bu, P);                // row below
}
template <int MODE>
__device__ __forceinline__ void grid_write(const SlotVal (&e)[3], const StageCfg& cfg, const GridCtx& gx, int reg, int v, int H,
                                           float* tile, const float* lm) {
    const StripPos p = strip_pos(gx.sg, v, H);
    if (gx.sg.ns == 2) {
        const EdgeSlot& es = gx.col[reg];
        edge_write<MODE>(e[0], cfg, es.pk, p.c ? 0 : WP - 1, es.aux < p.hv, tile, lm);
    }
    const EdgeSlot& rs = gx.row[reg];
    const int cc = p.c + rs.aux;
    const bool colok = cc >= 0 && cc < gx.sg.ns;
    edge_write<MODE>(e[1], cfg, rs.pk, 0, colok && p.r > 0, tile, lm);
    edge_write<MODE>(e[2], cfg, rs.pk, (H + 1) * WP, colok && p.r < gx.sg.nr - 1, tile, lm);
}

struct ConvEpilogue {
    const float* res;
    float* out;
    const float* xs;
    float xshift, xrstd;   // xhat = |xs| * xrstd + xshift (xshift = -mean * rstd)
    int cout, P;
    bool cvalid;
    bool xadd;       // data gradient, no statistics wanted (xs_stats == nullptr): xs is a tensor to ADD to the output (layer 1: the
                     // skip gradient ds_2 joins dx_0 here, so conv0's weight gradient reads one map instead of two)
    float* pool;     // forward, last layer: [B][npg][48] sums of |out| over a wave's positions (npg = 4 * slices position groups
    int npg, pg;     // per utterance; this wave is group pg) -- the head's spatial mean without a second pass over s_6
};

// Epilogue of one utterance for one wave: lane holds cout = 16nt + (lane&15) and, for tile j = t0 + ts * i, positions
// 16j + 4*(lane>>4) + {0,1,2,3}.  Its own operands (residual / saved activation at the output positions) are requested before
// the last K segment (conv_loop) so that their HBM latency is not exposed.
template <int NTW, int TS>
struct EpiAddr {
    unsigned boff, bmax;
    static constexpr unsigned tstep = 64u * TS;     // bytes between two tiles of this wave
    __device__ __forceinline__ EpiAddr(const ConvEpilogue& e, int t0, int lane) {
        // uniform base + one 32-bit lane offset + immediates; only a wave's last tile can overhang P (clamped)
        const unsigned crow = (unsigned)((e.cvalid ? e.cout : NMAP - 1) * e.P);
        boff = 4u * (crow + 16u * t0 + 4u * (lane >> 4));
        bmax = 4u * (crow + e.P - 2);
    }
    __device__ __forceinline__ unsigned off(int i, int hh) const {
        unsigned o = boff + tstep * i + 8u * hh;
        if (i == NTW - 1) o = o < bmax ? o : bmax;
        return o;
    }
};

template <int MODE, int NTW, int TS, int HALO = 0>
__device__ __forceinline__ void conv_epilogue(const f32x4 (&acc)[NTW], const float2 (&ev)[NTW > 0 ? NTW : 1][2], const ConvEpilogue& e,
                                              size_t ubase, int t0, int lane, float& st0, float& st1, int b,
                                              int pv = 0 /* HALO == 2: positions >= pv are outside the image (no part in the sums) */) {
    const EpiAddr<NTW, TS> ea(e, t0, lane);
    constexpr int ts = TS;
    char* obase = reinterpret_cast<char*>(e.out + ubase);
    float u0 = 0.0f, u1 = 0.0f;      // this utterance's share of the two statistics sums
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int mbase = 16 * (t0 + ts * i) + 4 * (lane >> 4);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int m = mbase + 2 * hh;
            if (e.cvalid && (i < NTW - 1 || m < e.P)) {
                float v0 = acc[i][2 * hh], v1 = acc[i][2 * hh + 1];
                if (MODE == 0) {
                    v0 = fmaxf(v0, 0.0f);
                    v1 = fmaxf(v1, 0.0f);
                    if (e.res != nullptr) {
                        // s = relu(conv) + skip >= 0; the ReLU mask of THIS convolution rides in the sign bit of the
                        // stored value (relu > 0 implies s > 0, so -s is a proper negative number); readers take |s|
                        const float2 r = ev[i][hh];
                        const bool k0 = v0 > 0.0f, k1 = v1 > 0.0f;
                        v0 += fabsf(r.x);
                        v1 += fabsf(r.y);
                        if (HALO != 2 || m < pv) {
                            u0 += v0 + v1;
                            u1 += v0 * v0 + v1 * v1;
                        }
                        v0 = k0 ? -v0 : v0;
                        v1 = k1 ? -v1 : v1;
                    } else if (HALO != 2 || m < pv) {
                        u0 += v0 + v1;
                        u1 += v0 * v0 + v1 * v1;
                    }
                } else if (e.xs != nullptr) {
                    const float2 sv = ev[i][hh];
                    if (e.xadd) {
                        v0 += sv.x;
                        v1 += sv.y;
                    } else if (HALO != 2 || m < pv) {
                        u0 += v0 + v1;
                        u1 += v0 * fmaf(fabsf(sv.x), e.xrstd, e.xshift) + v1 * fmaf(fabsf(sv.y), e.xrstd, e.xshift);
                    }
                }
                *reinterpret_cast<float2*>(obase + (ea.boff + ea.tstep * i + 8u * hh)) = make_float2(v0, v1);
            }
        }
    }
    st0 += u0;
    st1 += u1;
    if (MODE == 0 && e.pool != nullptr) {
        // last layer: the head's spatial mean without a second pass over s_6 -- u0 is the sum of the (non-negative) outputs over
        // this wave's positions; fold the four lane groups (lanes l, l^16, l^32, l^48 hold the same cout) and leave one value
        // per (utterance, position group, cout)
        float up = u0;
        up += __shfl_xor(up, 16);
        up += __shfl_xor(up, 32);
        if (lane < 16 && e.cvalid) e.pool[((size_t)b * e.npg + e.pg) * CP + e.cout] = up;
    }
}

struct ConvLoop {
    const lds_f32* ltile;
    const lds_f32* wnt;
    float* tile;
    const float* lm;
    int B, CS, t0, lane, tid;
    int nblk;   // utterance strides of the batch loop: workgroups (or groups of `slices` workgroups) sharing the batch
};


// All utterances b, b + nblk, ... of this workgroup; on entry channels 0..23 of utterance b are in the tile (barrier passed).
// Instantiated once per tile count (waves of one workgroup run different instances; every instance executes the same two
// barriers per utterance): the register allocator then sees one variant's live values, not the union of all five.
template <int MODE, int NTW, int TS, int HALO = 0>
__device__ __forceinline__ void conv_loop(const ConvLoop& c, const ConvEpilogue& epi, const StageCfg& cfg, const int (&pk0)[NS0],
                                          const int (&pk1)[NS1], int b, float& st0, float& st1, int& pslot,
                                          const HaloSlot (&hs)[2] = {HaloSlot{-1, 0}, HaloSlot{-1, 0}},
                                          const GridCtx& gx = GridCtx{}) {
    const int P = epi.P, tid = c.tid, lane = c.lane;
    const int wave = tid >> 6;
    const size_t r1 = (size_t)SPLIT_C * P;
    const float* eop = (MODE == 0) ? epi.res : epi.xs;
    constexpr int NTV = NTW > 0 ? NTW : 1;
    int dl[3];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
        const int tap = min(4 * s3 + (lane >> 4), 8);   // taps 9..11 meet zero weights: any finite value will do
        dl[s3] = (tap / 3) * WP + tap % 3 - (lane >> 4) * c.CS;
    }
    // HALO: strip parity of this workgroup's utterances (nblk is even) -> which edge column is fetched, and where it goes
    const int gcol = (b & 1) ? PW - 1 : 0, lcol = (b & 1) ? 0 : WP - 1;
    for (; b < c.B; b += c.nblk) {
        const size_t ubase = (size_t)b * NMAP * P;
        const int bn = b + c.nblk;
        const bool more = bn < c.B;                       // (uniform)
        const size_t nbase = (size_t)bn * NMAP * P;
        SlotVal v[3];
        // HALO == 2: three edge values per region (column halo, row above, row below), the utterance index for the pooled
        // gradient's broadcast, and whether this strip / the next one is a last row strip (rows >= hv_last: zeros)
        SlotVal ge[3];
        int bsl = b, bnsl = bn, pv = P;
        bool zb = false, zn = false;
        if constexpr (HALO == 2) {
            const StripPos pb = strip_pos(gx.sg, b, P / PW), pn = strip_pos(gx.sg, more ? bn : b, P / PW);
            bsl = pb.bu, bnsl = pn.bu, pv = pb.hv * PW;
            zb = pb.hv < P / PW, zn = pn.hv < P / PW;
        }
        f32x4 acc[NTV];
        KCursor<NTV> k;
        if constexpr (NTW > 0) k_begin<NTW, TS>(k, acc, c.ltile, c.wnt, c.CS, P, c.t0, lane);
        if constexpr (NTW > 0) k_prime<NTW>(k);
        // ---- phase A: channels 0..23 feed the matrix pipe, channels 24..44 of this utterance arrive
        constexpr bool STAGE = true;
        const int a0 = 2, a1 = 2, a2 = KG_A - 4, b0 = 2, b1 = 2, b2 = KG_B - 4;
#define HOWL_WINO_CHUNK() ((void)0)
#define HOWL_WINO_BAR() ((void)0)
        if constexpr (STAGE) {
            slot_load<MODE, HALO>(v[0], cfg, ubase + r1, tid, pk1[0], bsl);
            slot_load<MODE, HALO>(v[1], cfg, ubase + r1, tid + CONV_THREADS, pk1[1], bsl);
            if constexpr (HALO == 1) halo_load<MODE>(v[2], cfg, (size_t)(b ^ 1) * NMAP * P, hs[1], gcol, b);
            if constexpr (HALO == 2) grid_load<MODE>(ge, cfg, gx, 1, b, P / PW, P);
        }
        HOWL_STAIR(3);
        HOWL_WINO_CHUNK();
        if constexpr (NTW > 0) k_run<NTW>(k, acc, c.CS, a0);
        HOWL_WINO_BAR();
        if constexpr (STAGE) {
            slot_write<MODE>(v[0], cfg, ubase + r1, tid, pk1[0], c.tile, c.lm, HALO == 2 && zb && ((gx.pad[1] >> 0) & 1));
            slot_write<MODE>(v[1], cfg, ubase + r1, tid + CONV_THREADS, pk1[1], c.tile, c.lm, HALO == 2 && zb && ((gx.pad[1] >> 1) & 1));
            if constexpr (HALO == 1) halo_write<MODE>(v[2], cfg, hs[1], lcol, c.tile, c.lm);
            if constexpr (HALO == 2) grid_write<MODE>(ge, cfg, gx, 1, b, P / PW, c.tile, c.lm);
            slot_load<MODE, HALO>(v[0], cfg, ubase + r1, tid + 2 * CONV_THREADS, pk1[2], bsl);
            slot_load<MODE, HALO>(v[1], cfg, ubase + r1, tid + 3 * CONV_THREADS, pk1[3], bsl);
        }
        HOWL_STAIR(2);
        HOWL_WINO_CHUNK();
        if constexpr (NTW > 0) k_run<NTW>(k, acc, c.CS, a1);
        HOWL_WINO_BAR();
        if constexpr (STAGE) {
            slot_write<MODE>(v[0], cfg, ubase + r1, tid + 2 * CONV_THREADS, pk1[2], c.tile, c.lm, HALO == 2 && zb && ((gx.pad[1] >> 2) & 1));
            slot_write<MODE>(v[1], cfg, ubase + r1, tid + 3 * CONV_THREADS, pk1[3], c.tile, c.lm, HALO == 2 && zb && ((gx.pad[1] >> 3) & 1));
        }
        HOWL_STAIR(1);
        HOWL_WINO_CHUNK();
        if constexpr (NTW > 0) k_run<NTW>(k, acc, c.CS, a2);
        HOWL_WINO_BAR();
        HOWL_STAIR(0);
        HOWL_PROBE(cfg, wave, lane, pslot++);   // phase A done
        __syncthreads();      // channels 24..44 complete; every wave is past its reads of channels 0..23
        HOWL_PROBE(cfg, wave, lane, pslot++);   // barrier
        // ---- phase B: channels 24..44 feed the matrix pipe, channels 0..23 of the NEXT utterance arrive
        if constexpr (NTW > 0) k_prime<NTW>(k);      // (what the last step of phase A requested ahead predates the barrier)
        if (STAGE && more) {
            slot_load<MODE, HALO>(v[0], cfg, nbase, tid, pk0[0], bnsl);
            slot_load<MODE, HALO>(v[1], cfg, nbase, tid + CONV_THREADS, pk0[1], bnsl);
            slot_load<MODE, HALO>(v[2], cfg, nbase, tid + 2 * CONV_THREADS, pk0[2], bnsl);
            if constexpr (HALO == 2) grid_load<MODE>(ge, cfg, gx, 0, bn, P / PW, P);
        }
        HOWL_STAIR(3);
        HOWL_WINO_CHUNK();
        if constexpr (NTW > 0) k_run<NTW>(k, acc, c.CS, b0);
        HOWL_WINO_BAR();
        if (STAGE && more) {
            slot_write<MODE>(v[0], cfg, nbase, tid, pk0[0], c.tile, c.lm, HALO == 2 && zn && ((gx.pad[0] >> 0) & 1));
            slot_write<MODE>(v[1], cfg, nbase, tid + CONV_THREADS, pk0[1], c.tile, c.lm, HALO == 2 && zn && ((gx.pad[0] >> 1) & 1));
            slot_write<MODE>(v[2], cfg, nbase, tid + 2 * CONV_THREADS, pk0[2], c.tile, c.lm, HALO == 2 && zn && ((gx.pad[0] >> 2) & 1));
            if constexpr (HALO == 2) grid_write<MODE>(ge, cfg, gx, 0, bn, P / PW, c.tile, c.lm);
            slot_load<MODE, HALO>(v[0], cfg, nbase, tid + 3 * CONV_THREADS, pk0[3], bnsl);
            slot_load<MODE, HALO>(v[1], cfg, nbase, tid + 4 * CONV_THREADS, pk0[4], bnsl);
            if constexpr (HALO == 1) halo_load<MODE>(v[2], cfg, (size_t)(bn ^ 1) * NMAP * P, hs[0], gcol, bn);
        }
        HOWL_STAIR(2);
        HOWL_WINO_CHUNK();
        if constexpr (NTW > 0) k_run<NTW>(k, acc, c.CS, b1);
        HOWL_WINO_BAR();
        if (STAGE && more) {
            slot_write<MODE>(v[0], cfg, nbase, tid + 3 * CONV_THREADS, pk0[3], c.tile, c.lm, HALO == 2 && zn && ((gx.pad[0] >> 3) & 1));
            slot_write<MODE>(v[1], cfg, nbase, tid + 4 * CONV_THREADS, pk0[4], c.tile, c.lm, HALO == 2 && zn && ((gx.pad[0] >> 4) & 1));
            if constexpr (HALO == 1) halo_write<MODE>(v[2], cfg, hs[0], lcol, c.tile, c.lm);
        }
        // the epilogue's operands take the registers the staging slots just released; they land under the last K segment
        float2 ev[NTV][2];
        if constexpr (NTW > 0) {
            if (eop != nullptr) {
                const EpiAddr<NTW, TS> ea(epi, c.t0, lane);
                const char* ebase = reinterpret_cast<const char*>(eop + ubase);
#pragma unroll
                for (int i = 0; i < NTW; ++i)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) ev[i][hh] = *reinterpret_cast<const float2*>(ebase + ea.off(i, hh));
            }
            HOWL_STAIR(1);
            k_run<NTW>(k, acc, c.CS, b2);
            k_tail<NTW>(k, acc, dl);
            HOWL_STAIR(0);
        }
        HOWL_PROBE(cfg, wave, lane, pslot++);   // phase B done
        // the epilogue runs in front of the barrier: the waves of a SIMD leave the K loop a few hundred cycles apart, and an
        // early one's stores go out under the others' last MFMAs (behind the barrier all twelve epilogues ran with the matrix
        // pipe idle: +1.6 us per forward launch, tools/variants4.py)
        if constexpr (NTW > 0) conv_epilogue<MODE, NTW, TS, HALO>(acc, ev, epi, ubase, c.t0, lane, st0, st1, b, pv);
        HOWL_PROBE(cfg, wave, lane, pslot++);   // epilogue done
        __syncthreads();      // channels 0..23 of the next utterance complete; every wave is past its reads of 24..44
        HOWL_PROBE(cfg, wave, lane, pslot++);   // barrier
    }
}

// BatchNorm-backward means of layer i for the fused data / weight gradient staging: m1 = sum dx / N, m2 = sum dx * xhat / N,
// folded by every workgroup from the partials of the data gradient above it (wave w: channels 4w .. 4w+3; fold_part_column:
// the same bits in every workgroup) -- what bn_relu_bwd_kernel's prologue did -- or copied from the head's m12 (layer 6).
struct BwdFold {
    const float* stats;   // {mean, rstd} of layer i
    const float* m12;     // ready-made means, or
    const float* part;    // partials [2][48][part_stride(nparts)]
    int nparts;
    double count;
};
// Leaves bn_relu_bwd_pair's constants in lm = [A | Bc | Cc][48]: A = rstd, Bc = -rstd^2 m2, Cc = rstd (mean rstd m2 - m1).
__device__ __forceinline__ void bwd_consts_to_lds(float* lm, int ch, float mean, float rstd, float m1, float m2) {
    lm[ch] = rstd;
    lm[CP + ch] = -rstd * rstd * m2;
    lm[2 * CP + ch] = rstd * (mean * rstd * m2 - m1);
}
__device__ __forceinline__ void bwd_fold_to_lds(float* lm, const BwdFold& f, int tid, int nthreads) {
    const int lane = tid & 63, wave = tid >> 6;
    if (f.part == nullptr) {
        if (tid < CP) bwd_consts_to_lds(lm, tid, f.stats[tid], f.stats[CP + tid], f.m12[tid], f.m12[CP + tid]);
    } else {
        const int c8 = lane >> 3;
        for (int ch0 = 4 * wave; ch0 < CP; ch0 += 4 * (nthreads >> 6)) {     // (12 waves: one trip)
            const int ch = ch0 + (c8 & 3);
            const float mean = f.stats[ch], rstd = f.stats[CP + ch];
            const double acc = fold_part_column(f.part, part_stride(f.nparts), f.nparts, (c8 < 4 ? 0 : CP) + ch, lane);
            const double second = __shfl_xor(acc, 32);     // column groups 0..3: sum dx, their partners 4..7: sum dx * xhat
            if (c8 < 4 && (lane & 7) == 0)
                bwd_consts_to_lds(lm, ch, mean, rstd, (float)(acc / f.count), (float)(second / f.count));
        }
    }
}

// The weight-gradient partials of the layer ABOVE (written by the previous pair launch, one row of [48][WNCOL] per weight-gradient
// workgroup) are folded by the data-gradient workgroups of this launch once their own work is done: the pair's duration is set
// by its weight-gradient role, the data-gradient role finishes ~10 us earlier, so the fold of 10.6 MB per layer costs nothing
// (it was one 18-us launch over all six layers at the end of the pass).  Fixed order: bit-reproducible.
struct WFold {
    const float* part;   // [nparts][48 * WNCOL], or nullptr
    int nparts;
    float* out;          // dW (45,45,3,3) of that layer
};

// MODE 0: forward   out = relu(conv(x)) [+ res]; stats = (sum, sumsq) of out per cout
// MODE 1: dgrad     out = conv(dz);              stats = (sum out, sum out * xhat) per cout, xhat from s_prev
template <int MODE, int SLICES, int HALO = 0>
__device__ __forceinline__ void conv3x3_body(
    StageCfg cfg,                         // the input tile (see StageCfg)
    const float* __restrict__ in_stats,   // forward: {mean[48], rstd[48]} applied on load, or nullptr
    const float* __restrict__ wp,         // packed weights [3][102][64]
    const float* __restrict__ res,        // fwd: residual (B,45,P) or nullptr
    float* __restrict__ out,              // (B,45,P)
    const float* __restrict__ xs,         // dgrad: s_{i-1} for xhat, or nullptr (no stats)
    const float* __restrict__ xs_stats,   // dgrad: {mean, rstd} of layer i-1
    float* __restrict__ part,             // [nblk][2][48] partial statistics, or nullptr
    float* __restrict__ pool,             // forward, last layer: per-utterance sums for the head (ConvEpilogue::pool), or nullptr
    int B, int H, int bid, int nblk,      // utterances bid, bid + nblk, ... of this convolution
    int slice,                            // small batches: SLICES (1, 2, 4) workgroups share every utterance's position tiles
    const BnFold& fold,                   // forward: the input's BatchNorm statistics still as partials (or part == nullptr)
    const BwdFold& bfold,                 // fused data gradient: where m1 / m2 come from
    const WFold& wf = WFold{nullptr, 0, nullptr},     // data gradient: weight-gradient partials to fold at the end
    const StripGeom& sg = StripGeom{1, 1, 0}) {       // HALO == 2: how the utterances of this launch are strips of longer maps
    HIP_DYNAMIC_SHARED(float, lds)
    const int P = H * PW;
    const int CS = chan_stride(H, HALO == 2);
    const int TF = tile_floats(H, HALO == 2);
    float* wl = lds;                       // [3][102][64] weight fragments
    float* tile = lds + 3 * KSTEPS * 64;   // one utterance's zero-haloed input map
    float* lm = tile + TF;                 // forward: [-mean * rstd | rstd][48]; data gradient: [A | Bc | Cc][48] (bn_relu_bwd_pair)
    float* red = lm + 4 * CP;              // [12][2][16]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nt = __builtin_amdgcn_readfirstlane(wave % 3);
    const int mg = __builtin_amdgcn_readfirstlane(wave / 3);
    const int ntiles = (P + 15) / 16;
    // With fewer utterances than CUs, `slices` workgroups take an utterance each: all of them stage the whole map (the 3x3
    // neighbourhoods need it; the copies come from L2) and the weights, and split the position tiles -- workgroup `slice`
    // owns tiles t0 + ts * i with t0 = mg + 4 * slice, ts = 4 * slices -- so the K loop, which is what a launch of 64
    // one-utterance workgroups spends its time in, shrinks by `slices`.  Statistics partials get one row per workgroup.
    // (SLICES is a template parameter: the tile stride sits in immediates of the epilogue's addresses, and the kernels run at
    // the register limit of three waves per SIMD)
    constexpr int slices = SLICES, ts = 4 * SLICES;
    const int t0 = mg + 4 * slice;
    const int ntw = t0 < ntiles ? (ntiles - t0 + ts - 1) / ts : 0;  // wave-uniform, <= 5
    const bool folding = MODE == 0 && fold.part != nullptr;
    cfg.affine = MODE == 0 && (in_stats != nullptr || folding);
    if (slice != 0) cfg.ds = nullptr;      // one writer per utterance

    int pk0[NS0], pk1[NS1];
    GridCtx gx{};
    HaloSlot hs[2] = {HaloSlot{-1, 0}, HaloSlot{-1, 0}};
    if constexpr (HALO == 2) {
        gx.sg = sg;
        gx.pad[0] = gx.pad[1] = 0u;
        region_slots_pad<NS0>(pk0, 0, SPLIT_C, P, CS, tid, sg.hv_last, gx.pad[0]);
        region_slots_pad<NS1>(pk1, SPLIT_C, NMAP - SPLIT_C, P, CS, tid, sg.hv_last, gx.pad[1]);
        gx.col[0] = col_edge_slot(0, SPLIT_C, H, P, CS, tid);
        gx.col[1] = col_edge_slot(SPLIT_C, NMAP - SPLIT_C, H, P, CS, tid);
        gx.row[0] = row_edge_slot(0, SPLIT_C, P, CS, tid);
        gx.row[1] = row_edge_slot(SPLIT_C, NMAP - SPLIT_C, P, CS, tid);
    } else {
        region_slots<NS0>(pk0, 0, SPLIT_C, P, CS, tid);
        region_slots<NS1>(pk1, SPLIT_C, NMAP - SPLIT_C, P, CS, tid);
    }
    if constexpr (HALO == 1) {
        hs[0] = halo_slot(0, SPLIT_C, H, P, CS, tid);
        hs[1] = halo_slot(SPLIT_C, NMAP - SPLIT_C, H, P, CS, tid);
    }

    // channels 0..23 of the first utterance are requested before anything else so that HBM latency overlaps the setup
    int b = bid;
    int pslot = 0;
    HOWL_PROBE(cfg, wave, lane, pslot++);   // entry
    SlotVal first[NS0];
    if (b < B) {
#pragma unroll
        for (int j = 0; j < NS0; ++j)
            slot_load<MODE, HALO>(first[j], cfg, (size_t)b * NMAP * P, tid + j * CONV_THREADS, pk0[j],
                                  HALO == 2 ? strip_pos(sg, b, H).bu : b);
    }
    SlotVal firsth;
    SlotVal firstg[3];
    if constexpr (HALO == 1) {
        if (b < B) halo_load<MODE>(firsth, cfg, (size_t)(b ^ 1) * NMAP * P, hs[0], (b & 1) ? PW - 1 : 0, b);
    }
    if constexpr (HALO == 2) {
        if (b < B) grid_load<MODE>(firstg, cfg, gx, 0, b, H, P);
    }
    {
        float4 wv[7];  // 3*102*16 float4 = 4896 <= 7 * 768: all loads in flight, then the LDS stores
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int i = tid + j * CONV_THREADS;
            wv[j] = (i < 3 * KSTEPS * 16) ? reinterpret_cast<const float4*>(wp)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        HOWL_PROBE(cfg, wave, lane, pslot++);   // first tile + weights requested
        if (MODE == 0 && folding) {
            // Column sums of the producer's partials while the weight loads are in flight, wave by wave with no LDS
            // scratch and no barrier of their own: wave w owns channels 4w..4w+3 (column groups 0..3: the channels' sums,
            // 4..7: their sums of squares; fold_part_column), and the lanes that end up with a channel's two totals write
            // its mean / rstd straight to LDS for the setup barrier below.
            const int c8 = lane >> 3;
            const int ch = 4 * wave + (c8 & 3);
            const double acc = fold_part_column(fold.part, part_stride(fold.nparts), fold.nparts, (c8 < 4 ? 0 : CP) + ch, lane);
            const double sq = __shfl_xor(acc, 32);     // column groups 0..3 hold the sums, 4..7 the sums of squares
            if (c8 < 4 && (lane & 7) == 0) {
                const double mean = acc / fold.count;
                double var = sq / fold.count - mean * mean;
                var = var < 0.0 ? 0.0 : var;
                const float fm = (ch < NMAP) ? (float)mean : 0.0f;
                const float fr = (ch < NMAP) ? (float)(1.0 / sqrt(var + (double)BN_EPS)) : 0.0f;
                lm[ch] = -fm * fr;
                lm[CP + ch] = fr;
                if (bid == 0 && slice == 0) {   // one publisher: later readers (backward pass) and the running buffers (cnn.py:142)
                    fold.stats_out[ch] = fm;
                    fold.stats_out[CP + ch] = fr;
                    if (ch < NMAP && fold.bn.running_mean != nullptr) {
                        const double unbiased = fold.count > 1.0 ? var * fold.count / (fold.count - 1.0) : var;
                        fold.bn.running_mean[ch] = (1.0f - BN_MOMENTUM) * fold.bn.running_mean[ch] + BN_MOMENTUM * (float)mean;
                        fold.bn.running_var[ch] = (1.0f - BN_MOMENTUM) * fold.bn.running_var[ch] + BN_MOMENTUM * (float)unbiased;
                    }
                    if (ch == 0 && fold.bn.num_batches != nullptr) fold.bn.num_batches[0] += 1;
                }
            }
        }
        if (MODE == 1 && cfg.fused) bwd_fold_to_lds(lm, bfold, tid, CONV_THREADS);
        HOWL_PROBE(cfg, wave, lane, pslot++);   // statistics folded
        zero_lds(tile, TF, tid, CONV_THREADS);
        HOWL_PROBE(cfg, wave, lane, pslot++);   // tile zeroed
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int i = tid + j * CONV_THREADS;
            if (i < 3 * KSTEPS * 16) reinterpret_cast<float4*>(wl)[i] = wv[j];
        }
        HOWL_PROBE(cfg, wave, lane, pslot++);   // weights in LDS
    }
    if (MODE == 0 && !folding && tid < CP) {
        lm[tid] = cfg.affine ? -in_stats[tid] * in_stats[CP + tid] : 0.0f;     // slot_write: xhat = |s| * rstd + this
        lm[CP + tid] = cfg.affine ? in_stats[CP + tid] : 1.0f;
    }
    const int cout = 16 * nt + (lane & 15);
    const bool cvalid = cout < NMAP;
    float xshift = 0.0f, xrstd = 1.0f;
    if (MODE == 1 && xs != nullptr && xs_stats != nullptr && cvalid) {
        xrstd = xs_stats[CP + cout];
        xshift = -xs_stats[cout] * xrstd;
    }
    float st0 = 0.0f, st1 = 0.0f;
    const ConvEpilogue epi{res, out, xs, xshift, xrstd, cout, P, cvalid, MODE == 1 && xs != nullptr && xs_stats == nullptr,
                           pool, 4 * slices, t0};
    __syncthreads();  // weights, zero fill and the per-channel constants visible before the first stage
    HOWL_PROBE(cfg, wave, lane, pslot++);   // setup barrier passed
    if (b < B) {
#pragma unroll
        for (int j = 0; j < NS0; ++j)
            slot_write<MODE>(first[j], cfg, (size_t)b * NMAP * P, tid + j * CONV_THREADS, pk0[j], tile, lm,
                             HALO == 2 && strip_pos(sg, b, H).hv < H && ((gx.pad[0] >> j) & 1));
        if constexpr (HALO == 1) halo_write<MODE>(firsth, cfg, hs[0], (b & 1) ? 0 : WP - 1, tile, lm);
        if constexpr (HALO == 2) grid_write<MODE>(firstg, cfg, gx, 0, b, H, tile, lm);
    }
    __syncthreads();  // channels 0..23 of the first utterance in place
    HOWL_PROBE(cfg, wave, lane, pslot++);   // first half tile staged

    const ConvLoop cl{(const lds_f32*)tile, (const lds_f32*)wl + nt * KSTEPS * 64, tile, lm, B, CS, t0, lane, tid, nblk};
    switch (ntw) {
        case 5: conv_loop<MODE, 5, ts, HALO>(cl, epi, cfg, pk0, pk1, b, st0, st1, pslot, hs, gx); break;
        case 4: conv_loop<MODE, 4, ts, HALO>(cl, epi, cfg, pk0, pk1, b, st0, st1, pslot, hs, gx); break;
        case 3: conv_loop<MODE, 3, ts, HALO>(cl, epi, cfg, pk0, pk1, b, st0, st1, pslot, hs, gx); break;
        case 2: conv_loop<MODE, 2, ts, HALO>(cl, epi, cfg, pk0, pk1, b, st0, st1, pslot, hs, gx); break;
        case 1: conv_loop<MODE, 1, ts, HALO>(cl, epi, cfg, pk0, pk1, b, st0, st1, pslot, hs, gx); break;
        default: conv_loop<MODE, 0, ts, HALO>(cl, epi, cfg, pk0, pk1, b, st0, st1, pslot, hs, gx); break;
    }

    HOWL_PROBE(cfg, wave, lane, pslot++);   // all utterances done
    if (part != nullptr) {
        // lanes l, l^16, l^32, l^48 hold the same cout: fold them, then fold the 4 position groups via LDS
        st0 += __shfl_xor(st0, 16);
        st0 += __shfl_xor(st0, 32);
        st1 += __shfl_xor(st1, 16);
        st1 += __shfl_xor(st1, 32);
        if (lane < 16) {
            red[(wave * 2 + 0) * 16 + lane] = st0;
            red[(wave * 2 + 1) * 16 + lane] = st1;
        }
        __syncthreads();
        if (tid < 2 * CP) {
            const int which = tid / CP, c = tid - which * CP;
            const int t3 = c >> 4, cl = c & 15;
            float s = 0.0f;
#pragma unroll
            for (int g = 0; g < 4; ++g) s += red[((g * 3 + t3) * 2 + which) * 16 + cl];
            part[((size_t)which * CP + c) * part_stride(nblk * slices) + bid * slices + slice] = s;   // transposed: see fold_part_column
        }
    }
    if (MODE == 1 && wf.part != nullptr) {
        // this workgroup's share of the 48 * WNCOL columns, two at a time: 96 pair lanes x 8 row groups, up to 16 rows in flight
        // per thread, the row groups combined through LDS (the tile is free: every wave is past the loop's last barrier)
        constexpr int NCOL2 = CP * WNCOL / 2;
        const int nwg = nblk * slices, wg = bid * slices + slice;
        const int per = (NCOL2 + nwg - 1) / nwg;
        const int q0 = wg * per, q1 = (q0 + per < NCOL2) ? q0 + per : NCOL2;
        const int pl = tid % 96, rg = tid / 96;
        float2* scratch = reinterpret_cast<float2*>(tile);
        for (int qb = q0; qb < q1; qb += 96) {       // (uniform trip count)
            const int q = qb + pl;
            float2 acc = make_float2(0.0f, 0.0f);
            if (q < q1) {
                const float2* src = reinterpret_cast<const float2*>(wf.part) + q;
                for (int g0 = rg; g0 < wf.nparts; g0 += 8 * 16) {
                    float2 v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int g = g0 + 8 * u;
                        v[u] = src[(size_t)(g < wf.nparts ? g : wf.nparts - 1) * NCOL2];      // clamped: all in flight
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (g0 + 8 * u < wf.nparts) {
                            acc.x += v[u].x;
                            acc.y += v[u].y;
                        }
                }
            }
            scratch[rg * 96 + pl] = acc;
            __syncthreads();
            if (rg == 0 && q < q1) {
                float2 tot = scratch[pl];
#pragma unroll
                for (int r = 1; r < 8; ++r) {
                    tot.x += scratch[r * 96 + pl].x;
                    tot.y += scratch[r * 96 + pl].y;
                }
                // column (cout, idx = 45 tap + cin) of the accumulator layout [48][WNCOL] -> dW[(cout * 45 + cin) * 9 + tap]
                const int col = 2 * q;
                const int co = col / WNCOL, i0 = col - co * WNCOL;
                if (co < NMAP && i0 < WTAPS) wf.out[(co * NMAP + i0 % NMAP) * 9 + i0 / NMAP] = tot.x;
                if (co < NMAP && i0 + 1 < WTAPS) wf.out[(co * NMAP + (i0 + 1) % NMAP) * 9 + (i0 + 1) / NMAP] = tot.y;
            }
            __syncthreads();
        }
    }
}

template <int MODE, int SLICES, int HALO = 0>
__global__ __launch_bounds__(CONV_THREADS) void conv3x3_mfma_kernel(StageCfg cfg, const float* __restrict__ in_stats,
                                                                    const float* __restrict__ wp,
                                                                    const float* __restrict__ res, float* __restrict__ out,
                                                                    const float* __restrict__ xs,
                                                                    const float* __restrict__ xs_stats,
                                                                    float* __restrict__ part, float* __restrict__ pool, int B,
                                                                    int H, int nblk, BnFold fold, BwdFold bfold, WFold wf,
                                                                    StripGeom sg) {
    // blocks x, x + 8, ... run on XCD x (the hardware deals blocks round-robin): the SLICES workgroups of an utterance
    // group sit on one XCD and share its L2 copy of the maps
    const int x = blockIdx.x & 7, y = blockIdx.x >> 3;
    const int slice = y % SLICES, bid = (y / SLICES) * 8 + x;
    if (bid >= nblk) return;
    conv3x3_body<MODE, SLICES, HALO>(cfg, in_stats, wp, res, out, xs, xs_stats, part, pool, B, H, bid, nblk, slice, fold, bfold, wf, sg);
}


// wgrad: dW[cout][cin][tap] += sum_{b,p} dz[b,cout,p] * x[b,cin,p + tap shift],  x = (s_prev - mean) * rstd
//
// One wave owns NB of the 27 N tiles (tap, cin tile) and all three cout tiles: 3*NB accumulator chains that live in
// registers across every utterance of the workgroup.
//
// Tiles (round 4; both CHANNEL-major for the reads: lane & 15 = channel, lane >> 4 = g = one of the four positions of a k-step,
// group g walks rows g, g + 4, ... one column per k-step, every operand address "lane base + immediate"):
//   tz  dz_i           [48][4R rows][pitch 11], no halo (read at the output positions only)
//   tx  x_{i-1}        [48][4R + 4 rows][pitch 13], zero-haloed, in two ROW regions: halo rows 0 .. 4R1+1 for the rounds
//                      0 .. R1-1 (phase 1), halo rows 4R1 .. 4R+1 for the rounds R1 .. R-1 (phase 2) stored two rows further
//                      down, so the two data rows both phases read exist twice and a region is only ever overwritten while
//                      the other phase runs:
//   phase 1 (rounds < R1)   -- meanwhile the BOTTOM rows of this utterance go to LDS (z rows >= 4R1, x region 2)
//   barrier
//   phase 2 (rounds >= R1)  -- meanwhile the TOP rows of the next utterance go to LDS (z rows < 4R1, x region 1)
//   barrier
// ds_read_b32 is served per 32-lane half over 32 banks; a half holds {channels 0..15} x {groups g, g+1}: conflict-free iff the
// channel stride is = 2 (mod 32) (all even banks) and the two groups' rows are an odd number of floats apart (pitches 11, 13).
// With HOWL_RES8_BWD_FUSED (default) dz_i is built on the way into LDS from (dx_i, s_i, dskip) exactly as the data gradient's
// staging does (slot_write<1>): neither role of the pair reads a dz tensor.
constexpr int WPZ = 11;
constexpr int WNT = 4, WNB = 5;   // staging slots per thread: top regions (<= 13 rows x 45 ch / 2 / 768), bottom regions (<= 16 rows)
__host__ __device__ inline int wgrad_r1(int H) { return wgrad_rounds(H) / 2; }
__host__ __device__ inline int chan_stride_z(int H) {
    const int cs = 4 * wgrad_rounds(H) * WPZ;
    return cs + (2 - (cs % 32) + 32) % 32;
}
__host__ __device__ inline int chan_stride_x(int H) {
    const int cs = (4 * wgrad_rounds(H) + 4) * WPW;
    return cs + (2 - (cs % 32) + 32) % 32;
}
__host__ __device__ inline int tile_floats_z(int H) { return CP * chan_stride_z(H) + 32; }
__host__ __device__ inline int tile_floats_x(int H) { return CP * chan_stride_x(H) + 96; }   // slack: the K loop requests one k-step past the end

template <int NB>
struct WCursor {
    const lds_f32* ap[3];    // dz rows of the three cout tiles, this lane's channel and position group
    const lds_f32* bp[NB];   // x rows of the N tiles (halo origin + tap shift)
    const lds_f32* ape;      // the extra chain (EX): dz rows of its cout tile, x rows of its N tile
    const lds_f32* bpe;
};

// `rounds` rows per position group = 10 k-steps each.  The operands of k-step s+1 are requested before the MFMAs of step
// s; all offsets inside a round are immediates (next column: +1 float, next row of this group: +4 rows).
template <int NB, bool EX>
__device__ __forceinline__ void wgrad_k_run(WCursor<NB>& c, f32x4 (&acc)[NB][3], f32x4& acce, float (&az)[3], float (&bx)[NB],
                                            float& aze, float& bxe, int rounds) {
#pragma nounroll
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int w = 0; w < PW; ++w) {
            const int noz = (w + 1 < PW) ? (w + 1) : 4 * WPZ;
            const int nox = (w + 1 < PW) ? (w + 1) : 4 * WPW;
            float nz[3], nx[NB], nze = 0.0f, nxe = 0.0f;
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) nz[mt] = c.ap[mt][noz];
#pragma unroll
            for (int i = 0; i < NB; ++i) nx[i] = c.bp[i][nox];
            if constexpr (EX) {
                nze = c.ape[noz];
                nxe = c.bpe[nox];
            }
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) {
                    acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(az[mt], bx[i], acc[i][mt], 0, 0, 0);
                }
            if constexpr (EX) acce = __builtin_amdgcn_mfma_f32_16x16x4f32(aze, bxe, acce, 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) az[mt] = nz[mt];
#pragma unroll
            for (int i = 0; i < NB; ++i) bx[i] = nx[i];
            if constexpr (EX) {
                aze = nze;
                bxe = nxe;
            }
        }
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) c.ap[mt] += 4 * WPZ;
#pragma unroll
        for (int i = 0; i < NB; ++i) c.bp[i] += 4 * WPW;
        if constexpr (EX) {
            c.ape += 4 * WPZ;
            c.bpe += 4 * WPW;
        }
    }
}

// Slot j of a ROW region (data rows h0 .. h1-1 of all 45 channels) moves float2 number tid + 768 j of the region:
// packed descriptor = LDS float offset (bits 0..14) | float offset inside the utterance's (45, P) map (bits 15..28); -1 = none.
// The slot's channel goes to byte `cbyte` of ch[j] (the z and the x slot of a pair share one register: the per-channel constants
// are then one bit-field extract away instead of a division of the map offset by P).
template <int NSL>
__device__ __forceinline__ void row_region_slots(int (&pk)[NSL], int (&ch)[NSL], int cbyte, int h0, int h1, int P, int CS, int pitch,
                                                 int row0, int col0, int tid) {
    const int nper = PW * (h1 > h0 ? h1 - h0 : 0);
    const int nsafe = nper > 0 ? nper : 1;
    const float inv = 1.0f / (float)nsafe;
#pragma unroll
    for (int j = 0; j < NSL; ++j) {
        const int e = 2 * (tid + j * CONV_THREADS);
        const int c = (int)(((float)e + 0.5f) * inv);          // floor(e / nsafe), exact (see region_slots)
        const int q = e - c * nsafe;
        const int hh = (int)(((float)q + 0.5f) * 0.1f);
        const int w = q - hh * PW;
        const int h = h0 + hh;
        pk[j] = (e < NMAP * nper) ? ((c * CS + (h + row0) * pitch + w + col0) | ((c * P + h * PW + w) << 15)) : -1;
        ch[j] |= (e < NMAP * nper ? c : 0) << (8 * cbyte);
    }
}
// the same, plus bit j of `padbits` = slot j lies in a row >= hv (a row beyond the valid ones of a last row strip: StripGeom)
template <int NSL>
__device__ __forceinline__ void row_region_slots_pad(int (&pk)[NSL], int (&ch)[NSL], int cbyte, int h0, int h1, int P, int CS,
                                                     int pitch, int row0, int col0, int tid, int hv, unsigned& padbits) {
    row_region_slots<NSL>(pk, ch, cbyte, h0, h1, P, CS, pitch, row0, col0, tid);
    const int nper = PW * (h1 > h0 ? h1 - h0 : 0), nsafe = nper > 0 ? nper : 1;
#pragma unroll
    for (int j = 0; j < NSL; ++j) {
        const int e = 2 * (tid + j * CONV_THREADS);
        const int h = h0 + (e % nsafe) / PW;
        if (pk[j] >= 0 && h >= hv) padbits |= 1u << j;
    }
}

struct WStage {
    StageCfg z;           // dz_i: plain (z.a = dz) or fused (z.a = dx_i ...), see StageCfg
    const float* x;       // s_{i-1}
    bool xaffine;         // x = (|s| - mean) * rstd, else |s|
};

struct WSlot {
    SlotVal z;
    float2 x;
};

// z slot: the data gradient's staging arithmetic (slot_write<1>) with this kernel's addressing; addresses are "uniform base +
// 32-bit lane offset" recomputed from the packed descriptor where they are used
template <int HALO = 0>
__device__ __forceinline__ void wz_load(SlotVal& v, const StageCfg& cfg, size_t ubase, int pkj, int chj, int b) {
    HOWL_OPAQUE_V(pkj);
    HOWL_OPAQUE_V(chj);
    const unsigned g = pkj >= 0 ? (unsigned)(pkj >> 15) : 0u;
    const unsigned off = 4u * g;
    if (cfg.fused) {
        if (cfg.a != nullptr) {
            v.a = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(cfg.a + ubase) + off);
        } else {
            const unsigned c4 = 4u * (unsigned)(chj & 0xFF);
            const float gg = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(cfg.dpool + (size_t)(HALO == 1 ? b >> 1 : b) * CP) + c4) * cfg.invP;
            v.a = make_float2(gg, gg);
        }
        v.s = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(cfg.s + ubase) + off);
        v.k = cfg.k != nullptr ? *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(cfg.k + ubase) + off)
                               : make_float2(0.0f, 0.0f);
    } else {
        v.a = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(cfg.a + ubase) + off);
    }
}
__device__ __forceinline__ void wz_write(const SlotVal& v, const StageCfg& cfg, int pkj, int chj, float* tz, const float* lm,
                                         bool zero = false) {
    HOWL_OPAQUE_V(pkj);
    HOWL_OPAQUE_V(chj);
    if (pkj < 0) return;
    float v0 = v.a.x, v1 = v.a.y;
    if (cfg.fused) {
        const int c = chj & 0xFF;
        float2 ds, dz;
        bn_relu_bwd_pair(v, lm + c, cfg.even, ds, dz);
        v0 = dz.x;
        v1 = dz.y;
    }
    if (zero) v0 = v1 = 0.0f;
    float* d = tz + (pkj & 0x7FFF);
    d[0] = v0;
    d[1] = v1;
}
__device__ __forceinline__ void wx_load(float2& v, const float* x, size_t ubase, int pkj) {
    HOWL_OPAQUE_V(pkj);
    const unsigned off = pkj >= 0 ? 4u * (unsigned)(pkj >> 15) : 0u;
    v = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(x + ubase) + off);
}
__device__ __forceinline__ void wx_write(const float2& v, bool affine, int pkj, int chj, float* tx, const float* lm,
                                         bool zero = false) {
    HOWL_OPAQUE_V(pkj);
    HOWL_OPAQUE_V(chj);
    if (pkj < 0) return;
    float v0 = fabsf(v.x), v1 = fabsf(v.y);      // |s|: see conv_epilogue
    if (affine) {
        const int c = (chj >> 8) & 0xFF;
        const float sh = lm[4 * CP + c], r = lm[5 * CP + c];
        v0 = fmaf(v0, r, sh);
        v1 = fmaf(v1, r, sh);
    }
    if (zero) v0 = v1 = 0.0f;
    float* d = tx + (pkj & 0x7FFF);
    d[0] = v0;
    d[1] = v1;
}

// HALO (wide maps, see HaloSlot): the x tile's halo column on the neighbour's side, one element per thread and row region
struct WHalo {
    int pk;   // LDS float offset of tile column 0 of the element's row in tx; -1 = no element
    int g;    // float offset of column 0 of that row in an utterance's (45, P) map
    int c;    // channel
    int h;    // the element's row (HALO == 2: rows beyond a last row strip's valid ones stay zero)
};
__device__ __forceinline__ WHalo whalo_slot(int h0, int h1, int P, int CSX, int row0, int tid) {
    const int nper = h1 > h0 ? h1 - h0 : 0, nsafe = nper > 0 ? nper : 1;
    const int c = tid / nsafe, hh = tid - c * nsafe, h = h0 + hh;
    const bool ok = tid < NMAP * nper;
    return WHalo{ok ? c * CSX + (h + row0) * WPW : -1, ok ? c * P + h * PW : 0, ok ? c : 0, ok ? h : 0};
}
__device__ __forceinline__ float whalo_load(const float* x, size_t nbase, const WHalo& wh, int gcol) {
    if (wh.pk < 0) return 0.0f;     // (no element: also keeps the address inside the tensor when there is no neighbour strip)
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(x + nbase) + 4u * (unsigned)(wh.g + gcol));
}
__device__ __forceinline__ void whalo_write(float v, bool affine, const WHalo& wh, int lcol, float* tx, const float* lm,
                                            int hv = 1 << 20) {
    if (wh.pk < 0) return;
    float v0 = fabsf(v);
    if (affine) v0 = fmaf(v0, lm[5 * CP + wh.c], lm[4 * CP + wh.c]);
    tx[wh.pk + lcol] = wh.h < hv ? v0 : 0.0f;
}

// HALO == 2 (StripGeom): the x tile's halo ROWS -- twelve columns of the last row of the strip above (tile row 0 of region 1)
// and of the first row of the strip below (tile row H + 3 of region 2), corners from the diagonal neighbours -- one scalar per
// thread and side (45 x 12 = 540 <= 768); what the workgroup carries through its utterance loop
struct WRowSlot {
    int pk;   // LDS float offset in tx of the element in tile row 0; -1 = none
    int g;    // float offset inside the source block: channel * P + source column
    int c;    // channel
    int dc;   // column-strip delta of the source block
};
__device__ __forceinline__ WRowSlot wrow_slot(int P, int CSX, int tid) {
    const int c = tid / WP, w = tid - c * WP;
    const bool ok = tid < NMAP * WP;
    const int dc = w == 0 ? -1 : (w == WP - 1 ? 1 : 0);
    const int col = w == 0 ? PW - 1 : (w == WP - 1 ? 0 : w - 1);
    return WRowSlot{ok ? c * CSX + w : -1, ok ? c * P + col : 0, ok ? c : 0, ok ? dc : 0};
}
struct WGridCtx {
    StripGeom sg;
    WRowSlot row;
    unsigned pzt, pxt, pzb, pxb;     // pad bits of the four slot arrays (rows >= hv_last)
};
__device__ __forceinline__ float wrow_load(const float* x, const WGridCtx& gx, int v, int H, int P, bool below) {
    const StripPos p = strip_pos(gx.sg, v, H);
    const int cc = p.c + gx.row.dc;
    const bool valid = gx.row.pk >= 0 && cc >= 0 && cc < gx.sg.ns && (below ? p.r < gx.sg.nr - 1 : p.r > 0);
    const int blk = below ? v + gx.sg.ns + gx.row.dc : v - gx.sg.ns + gx.row.dc;
    const size_t idx = valid ? (size_t)blk * NMAP * P + (size_t)(gx.row.g + (below ? 0 : (H - 1) * PW)) : 0;
    return valid ? x[idx] : 0.0f;
}
__device__ __forceinline__ void wrow_write(float v, bool affine, const WGridCtx& gx, int vstrip, int H, bool below, float* tx,
                                           const float* lm) {
    if (gx.row.pk < 0) return;
    const StripPos p = strip_pos(gx.sg, vstrip, H);
    const int cc = p.c + gx.row.dc;
    const bool valid = cc >= 0 && cc < gx.sg.ns && (below ? p.r < gx.sg.nr - 1 : p.r > 0);
    float v0 = fabsf(v);
    if (affine) v0 = fmaf(v0, lm[5 * CP + gx.row.c], lm[4 * CP + gx.row.c]);
    tx[gx.row.pk + (below ? (H + 3) * WPW : 0)] = valid ? v0 : 0.0f;
}

struct WgradArgs {
    WStage st;
    float* part;
    float* tz;
    float* tx;
    const float* lm;
    int B, P, CSZ, CSX, R, R1, tid, lane, wave;
    int bid, nblk;   // utterances bid, bid + nblk, ...; partial row bid
    int gw;          // this wave's index among the GWS = 12 * slices waves that share the 26 N tiles of those utterances
    int qe, mte;     // EX: this wave's extra chain = (N tile qe, cout tile mte)
};

// all utterances b, b + nblk, ... of this workgroup (the top rows of utterance b are in the tiles on entry), then this wave's partials
// x-tile offset of this lane's column of N tile q: column idx = 16 q + n = 45 * tap + cin (the 11 columns past 405 read the
// all-zero channel 45)
__device__ __forceinline__ int wgrad_boff(int q, int n, int g, int CSX) {
    const int idx = 16 * q + n;
    const int tap = idx < WTAPS ? idx / NMAP : 0;
    const int cin = idx < WTAPS ? idx - tap * NMAP : NMAP;
    return cin * CSX + (g + tap / 3) * WPW + (tap % 3);     // cin row, halo origin + tap shift
}

template <int NB, bool EX, int GWS, int HALO = 0>
__device__ __forceinline__ void wgrad_loop(const WgradArgs& a, const int (&zt)[WNT], const int (&xt)[WNT], const int (&ct)[WNT],
                                           const int (&zb)[WNB], const int (&xb)[WNB], const int (&cb)[WNB], int b, int& pslot,
                                           const WHalo (&wh)[2] = {WHalo{-1, 0, 0, 0}, WHalo{-1, 0, 0, 0}},
                                           const WGridCtx& gx = WGridCtx{}) {
    const int lane = a.lane, wave = a.wave;
    const int g = lane >> 4, n = lane & 15;
    f32x4 acc[NB][3];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) acc[i][mt] = {0.0f, 0.0f, 0.0f, 0.0f};
    // N tiles q = gw, gw + GWS, ... (< 26)
    int boff[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) boff[i] = wgrad_boff(a.gw + GWS * i, n, g, a.CSX);
    const int boffe = EX ? wgrad_boff(a.qe, n, g, a.CSX) : 0;
    f32x4 acce = {0.0f, 0.0f, 0.0f, 0.0f};
    const int aoff = n * a.CSZ + g * WPZ;                                    // cout row g, column 0
    const int R1 = a.R1, R2 = a.R - a.R1;
    // HALO: tx columns 0 / 11 are the halo of the data columns 1..10; strip parity is fixed per workgroup (nblk is even)
    const int gcol = (b & 1) ? PW - 1 : 0, lcol = (b & 1) ? 0 : PW + 1;
    for (; b < a.B; b += a.nblk) {
        const size_t ubase = (size_t)b * NMAP * a.P;
        const int bn = b + a.nblk;
        const bool more = bn < a.B;                          // (uniform)
        const size_t nbase = (size_t)bn * NMAP * a.P;
        WCursor<NB> c;
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) c.ap[mt] = (const lds_f32*)a.tz + aoff + 16 * mt * a.CSZ;
#pragma unroll
        for (int i = 0; i < NB; ++i) c.bp[i] = (const lds_f32*)a.tx + boff[i];
        c.ape = (const lds_f32*)a.tz + aoff + 16 * a.mte * a.CSZ;
        c.bpe = (const lds_f32*)a.tx + boffe;
        float az[3], bx[NB], aze = 0.0f, bxe = 0.0f;
        WSlot v[2];
        // HALO == 2 (StripGeom): utterance index for the pooled gradient's broadcast, valid rows of this strip and the next one
        int bsl = b, bnsl = bn, hvb = 1 << 20, hvn = 1 << 20;
        bool zb_ = false, zn_ = false;
        if constexpr (HALO == 2) {
            const int Hs = a.P / PW;
            const StripPos pb = strip_pos(gx.sg, b, Hs), pn = strip_pos(gx.sg, more ? bn : b, Hs);
            bsl = pb.bu, bnsl = pn.bu, hvb = pb.hv, hvn = pn.hv;
            zb_ = pb.hv < Hs, zn_ = pn.hv < Hs;
        }
        // ---- phase 1: rounds 0 .. R1-1 on the top rows; the bottom rows of this utterance arrive
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) az[mt] = c.ap[mt][0];
#pragma unroll
        for (int i = 0; i < NB; ++i) bx[i] = c.bp[i][0];
        if constexpr (EX) {
            aze = c.ape[0];
            bxe = c.bpe[0];
        }
        static_assert(WNB == 5 && WNT == 4, "the staging schedule below is written for 5 + 4 slot pairs");
#define HOWL_W_LOAD(slot_, j_, zpk_, xpk_, cpk_, ub_, bb_)                    \
    do {                                                                     \
        wz_load<HALO>(v[slot_].z, a.st.z, ub_, zpk_[j_], cpk_[j_], bb_);     \
        wx_load(v[slot_].x, a.st.x, ub_, xpk_[j_]);                          \
    } while (0)
#define HOWL_W_WRITE(slot_, j_, zpk_, xpk_, cpk_, zf_, zp_, xp_)                                                             \
    do {                                                                                                                   \
        wz_write(v[slot_].z, a.st.z, zpk_[j_], cpk_[j_], a.tz, a.lm, HALO == 2 && (zf_) && (((zp_) >> (j_)) & 1));         \
        wx_write(v[slot_].x, a.st.xaffine, xpk_[j_], cpk_[j_], a.tx, a.lm, HALO == 2 && (zf_) && (((xp_) >> (j_)) & 1));   \
    } while (0)
        HOWL_W_LOAD(0, 0, zb, xb, cb, ubase, bsl);
        HOWL_W_LOAD(1, 1, zb, xb, cb, ubase, bsl);
        float xh = 0.0f, xr = 0.0f;
        if constexpr (HALO != 0) xh = whalo_load(a.st.x, (size_t)(b ^ 1) * NMAP * a.P, wh[1], gcol);
        if constexpr (HALO == 2) xr = wrow_load(a.st.x, gx, b, a.P / PW, a.P, true);
        HOWL_STAIR(3);
        if (R1 > 0) wgrad_k_run<NB, EX>(c, acc, acce, az, bx, aze, bxe, 1);
        HOWL_W_WRITE(0, 0, zb, xb, cb, zb_, gx.pzb, gx.pxb);
        HOWL_W_WRITE(1, 1, zb, xb, cb, zb_, gx.pzb, gx.pxb);
        if constexpr (HALO != 0) whalo_write(xh, a.st.xaffine, wh[1], lcol, a.tx, a.lm, hvb);
        if constexpr (HALO == 2) wrow_write(xr, a.st.xaffine, gx, b, a.P / PW, true, a.tx, a.lm);
        HOWL_W_LOAD(0, 2, zb, xb, cb, ubase, bsl);
        HOWL_W_LOAD(1, 3, zb, xb, cb, ubase, bsl);
        HOWL_STAIR(2);
        if (R1 > 1) wgrad_k_run<NB, EX>(c, acc, acce, az, bx, aze, bxe, 1);
        HOWL_W_WRITE(0, 2, zb, xb, cb, zb_, gx.pzb, gx.pxb);
        HOWL_W_WRITE(1, 3, zb, xb, cb, zb_, gx.pzb, gx.pxb);
        HOWL_W_LOAD(0, 4, zb, xb, cb, ubase, bsl);
        HOWL_STAIR(1);
        if (R1 > 2) wgrad_k_run<NB, EX>(c, acc, acce, az, bx, aze, bxe, R1 - 2);
        HOWL_STAIR(0);
        HOWL_W_WRITE(0, 4, zb, xb, cb, zb_, gx.pzb, gx.pxb);
        HOWL_PROBE(a.st.z, wave, lane, pslot++);   // phase 1 done
        __syncthreads();      // bottom rows complete; every wave is past its reads of the top rows
        HOWL_PROBE(a.st.z, wave, lane, pslot++);   // barrier
        // ---- phase 2: rounds R1 .. R-1 on the bottom rows (x: region 2, two tile rows further down); the top rows of the
        // next utterance arrive.  The operands requested ahead by the last k-step of phase 1 predate the barrier: re-read.
#pragma unroll
        for (int i = 0; i < NB; ++i) c.bp[i] += 2 * WPW;
        c.bpe += 2 * WPW;
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) az[mt] = c.ap[mt][0];
#pragma unroll
        for (int i = 0; i < NB; ++i) bx[i] = c.bp[i][0];
        if constexpr (EX) {
            aze = c.ape[0];
            bxe = c.bpe[0];
        }
        if (more) {
            HOWL_W_LOAD(0, 0, zt, xt, ct, nbase, bnsl);
            HOWL_W_LOAD(1, 1, zt, xt, ct, nbase, bnsl);
            if constexpr (HALO != 0) xh = whalo_load(a.st.x, (size_t)(bn ^ 1) * NMAP * a.P, wh[0], gcol);
            if constexpr (HALO == 2) xr = wrow_load(a.st.x, gx, bn, a.P / PW, a.P, false);
        }
        HOWL_STAIR(3);
        wgrad_k_run<NB, EX>(c, acc, acce, az, bx, aze, bxe, 1);
        if (more) {
            HOWL_W_WRITE(0, 0, zt, xt, ct, zn_, gx.pzt, gx.pxt);
            HOWL_W_WRITE(1, 1, zt, xt, ct, zn_, gx.pzt, gx.pxt);
            if constexpr (HALO != 0) whalo_write(xh, a.st.xaffine, wh[0], lcol, a.tx, a.lm, hvn);
            if constexpr (HALO == 2) wrow_write(xr, a.st.xaffine, gx, bn, a.P / PW, false, a.tx, a.lm);
            HOWL_W_LOAD(0, 2, zt, xt, ct, nbase, bnsl);
            HOWL_W_LOAD(1, 3, zt, xt, ct, nbase, bnsl);
        }
        HOWL_STAIR(2);
        if (R2 > 1) wgrad_k_run<NB, EX>(c, acc, acce, az, bx, aze, bxe, 1);
        if (more) {
            HOWL_W_WRITE(0, 2, zt, xt, ct, zn_, gx.pzt, gx.pxt);
            HOWL_W_WRITE(1, 3, zt, xt, ct, zn_, gx.pzt, gx.pxt);
        }
        HOWL_STAIR(1);
        if (R2 > 2) wgrad_k_run<NB, EX>(c, acc, acce, az, bx, aze, bxe, R2 - 2);
        HOWL_STAIR(0);
#undef HOWL_W_LOAD
#undef HOWL_W_WRITE
        HOWL_PROBE(a.st.z, wave, lane, pslot++);   // phase 2 done
        __syncthreads();      // top rows of the next utterance complete; every wave is past its reads of the bottom rows
        HOWL_PROBE(a.st.z, wave, lane, pslot++);   // barrier
    }
    // D[row = cout = 16mt + 4*(lane>>4) + r][col = n = lane&15] of N tile q -> partial row [48][WNCOL], column 16 q + n
    float* dst = a.part + (size_t)a.bid * CP * WNCOL;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int q = a.gw + GWS * i;
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(16 * mt + 4 * g + r) * WNCOL + 16 * q + n] = acc[i][mt][r];
    }
    if constexpr (EX) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(16 * a.mte + 4 * g + r) * WNCOL + 16 * a.qe + n] = acce[r];
    }
}

template <int SLICES, int HALO = 0>
__device__ __forceinline__ void wgrad_body(
    WStage st, const float* __restrict__ in_stats /* {mean, rstd} of layer i-1 or nullptr */, const BwdFold& bfold,
    float* __restrict__ part /* [nblk][48][WNCOL] */, int B, int H, int bid, int nblk, int slice,
    const StripGeom& sg = StripGeom{1, 1, 0} /* HALO == 2 */) {
    HIP_DYNAMIC_SHARED(float, lds)
    const int P = H * PW;
    const int CSZ = chan_stride_z(H), CSX = chan_stride_x(H);
    float* tz = lds;
    float* tx = lds + tile_floats_z(H);
    float* lm = tx + tile_floats_x(H);      // [A | Bc | Cc (bn_relu_bwd_pair) | - | -mean_{i-1} rstd_{i-1} | rstd_{i-1}][48]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int R = wgrad_rounds(H), R1 = wgrad_r1(H);
    st.xaffine = in_stats != nullptr;
    st.z.ds = nullptr;                      // ds_i is written by the data gradient's staging
    int pslot = 0;
    HOWL_PROBE(st.z, wave, lane, pslot++);   // entry

    // row regions: phase 1 reads z rows < 4 R1 and x halo rows <= 4 R1 + 1 (data rows <= 4 R1); phase 2 the z rows >= 4 R1 and
    // x halo rows >= 4 R1 (data rows >= 4 R1 - 1), kept two tile rows further down
    const int zsplit = 4 * R1 < H ? 4 * R1 : H;
    const int xtop = R1 > 0 ? (4 * R1 + 1 < H ? 4 * R1 + 1 : H) : 0;
    const int xbot = 4 * R1 - 1 > 0 ? 4 * R1 - 1 : 0;
    int zt[WNT], xt[WNT], zb[WNB], xb[WNB], ct[WNT] = {}, cb[WNB] = {};
    WGridCtx gx{};
    if constexpr (HALO == 2) {
        gx.sg = sg;
        gx.row = wrow_slot(P, CSX, tid);
        row_region_slots_pad<WNT>(zt, ct, 0, 0, zsplit, P, CSZ, WPZ, 0, 0, tid, sg.hv_last, gx.pzt);
        row_region_slots_pad<WNT>(xt, ct, 1, 0, xtop, P, CSX, WPW, 1, 1, tid, sg.hv_last, gx.pxt);
        row_region_slots_pad<WNB>(zb, cb, 0, zsplit, H, P, CSZ, WPZ, 0, 0, tid, sg.hv_last, gx.pzb);
        row_region_slots_pad<WNB>(xb, cb, 1, xbot, H, P, CSX, WPW, 3, 1, tid, sg.hv_last, gx.pxb);
    } else {
        row_region_slots<WNT>(zt, ct, 0, 0, zsplit, P, CSZ, WPZ, 0, 0, tid);
        row_region_slots<WNT>(xt, ct, 1, 0, xtop, P, CSX, WPW, 1, 1, tid);
        row_region_slots<WNB>(zb, cb, 0, zsplit, H, P, CSZ, WPZ, 0, 0, tid);
        row_region_slots<WNB>(xb, cb, 1, xbot, H, P, CSX, WPW, 3, 1, tid);
    }
    WHalo wh[2] = {WHalo{-1, 0, 0, 0}, WHalo{-1, 0, 0, 0}};
    if (HALO == 1 || (HALO == 2 && sg.ns == 2)) {      // the neighbour column strip's edge column
        wh[0] = whalo_slot(0, xtop, P, CSX, 1, tid);
        wh[1] = whalo_slot(xbot, H, P, CSX, 3, tid);
    }
    // the first utterance's top rows are requested before the LDS setup so that HBM latency overlaps it
    const int b = bid;
    WSlot first[WNT];
    if (b < B) {
#pragma unroll
        for (int j = 0; j < WNT; ++j) {
            wz_load<HALO>(first[j].z, st.z, (size_t)b * NMAP * P, zt[j], ct[j], HALO == 2 ? strip_pos(sg, b, H).bu : b);
            wx_load(first[j].x, st.x, (size_t)b * NMAP * P, xt[j]);
        }
    }
    float firsth = 0.0f, firstr = 0.0f;
    if constexpr (HALO != 0) {
        if (b < B) firsth = whalo_load(st.x, (size_t)(b ^ 1) * NMAP * P, wh[0], (b & 1) ? PW - 1 : 0);
    }
    if constexpr (HALO == 2) {
        if (b < B) firstr = wrow_load(st.x, gx, b, H, P, false);
    }
    zero_lds(lds, tile_floats_z(H) + tile_floats_x(H), tid, CONV_THREADS);
    if (st.z.fused) {
        bwd_fold_to_lds(lm, bfold, tid, CONV_THREADS);
    }
    if (tid < CP) {
        lm[4 * CP + tid] = st.xaffine ? -in_stats[tid] * in_stats[CP + tid] : 0.0f;    // wx_write: xhat = |s| * rstd + this
        lm[5 * CP + tid] = st.xaffine ? in_stats[CP + tid] : 1.0f;
    }
    __syncthreads();
    if (b < B) {
#pragma unroll
        for (int j = 0; j < WNT; ++j) {
            const bool padded = HALO == 2 && strip_pos(sg, b, H).hv < H;
            wz_write(first[j].z, st.z, zt[j], ct[j], tz, lm, padded && ((gx.pzt >> j) & 1));
            wx_write(first[j].x, st.xaffine, xt[j], ct[j], tx, lm, padded && ((gx.pxt >> j) & 1));
        }
        if constexpr (HALO != 0) whalo_write(firsth, st.xaffine, wh[0], (b & 1) ? 0 : PW + 1, tx, lm, HALO == 2 ? strip_pos(sg, b, H).hv : 1 << 20);
        if constexpr (HALO == 2) wrow_write(firstr, st.xaffine, gx, b, H, false, tx, lm);
    }
    __syncthreads();
    HOWL_PROBE(st.z, wave, lane, pslot++);   // prologue done
    // instantiated per tile count (waves 0..2 carry a third N tile): no branches inside the K loop, and the register
    // allocator sees one variant's live values (waves of a workgroup run different instances with the same barriers)
    // small batches: two workgroups share an utterance group's 27 N tiles (each stages both maps; wave gw of the 24 owns tiles
    // gw, gw + 24) and write disjoint columns of the same partial row
    constexpr int GWS = 12 * SLICES;
    const int gw = wave + 12 * slice;
    if constexpr (SLICES == 1) {
        // tiles wave, wave + 12 for everyone; the six chains of tiles 24 / 25 go to waves 0, 4 (SIMD 0), 1, 5 (SIMD 1), 2, 3
        int qe = 0, mte = 0;
        bool ex = true;
        switch (wave) {
            case 0: qe = 24, mte = 0; break;
            case 4: qe = 24, mte = 1; break;
            case 1: qe = 24, mte = 2; break;
            case 5: qe = 25, mte = 0; break;
            case 2: qe = 25, mte = 1; break;
            case 3: qe = 25, mte = 2; break;
            default: ex = false; break;
        }
        const WgradArgs a{st, part, tz, tx, lm, B, P, CSZ, CSX, R, R1, tid, lane, wave, bid, nblk, gw, qe, mte};
        if (ex)
            wgrad_loop<2, true, GWS, HALO>(a, zt, xt, ct, zb, xb, cb, b, pslot, wh, gx);
        else
            wgrad_loop<2, false, GWS, HALO>(a, zt, xt, ct, zb, xb, cb, b, pslot, wh, gx);
    } else {
        // small batches: tile gw for each of the 24 waves of the two workgroups; the six chains of tiles 24 / 25 (three cout tiles
        // each) go one each to waves 0, 1, 2 of either workgroup -- three different SIMDs (round 6: as whole tiles on waves 0 / 1 of
        // slice 0 they made that workgroup's SIMD 0 carry 840 MFMAs per utterance against 630 everywhere else, and the launch
        // waits for its slowest SIMD; the same chains in the same order: bit-identical partials)
        static_assert(SLICES == 2, "the weight gradient is sliced two ways");
        const bool ex = wave < 3;
        const WgradArgs a{st, part, tz, tx, lm, B, P, CSZ, CSX, R, R1, tid, lane, wave, bid, nblk, gw, 24 + slice, ex ? wave : 0};
        if (ex)
            wgrad_loop<1, true, GWS, HALO>(a, zt, xt, ct, zb, xb, cb, b, pslot, wh, gx);
        else
            wgrad_loop<1, false, GWS, HALO>(a, zt, xt, ct, zb, xb, cb, b, pslot, wh, gx);
    }
    HOWL_PROBE(st.z, wave, lane, pslot++);   // partials written
}

template <int SLICES, int HALO = 0>
__global__ __launch_bounds__(CONV_THREADS) void wgrad_mfma_kernel(WStage st, const float* __restrict__ in_stats, BwdFold bfold,
                                                                  float* __restrict__ part, int B, int H, int nblk, StripGeom sg) {
    const int x = blockIdx.x & 7, y = blockIdx.x >> 3;
    const int slice = y % SLICES, bid = (y / SLICES) * 8 + x;
    if (bid >= nblk) return;
    wgrad_body<SLICES, HALO>(st, in_stats, bfold, part, B, H, bid, nblk, slice, sg);
}

// Data gradient and weight gradient of one layer in ONE launch.  Both hang off dz_i and are independent; side by side on
// half the CUs each, a workgroup carries twice the utterances, so the per-launch costs (dispatch, weight / LDS prologue,
// first-tile fetch, tail) are paid once per two utterance passes.  (Round 1 ran them on two HIP queues: the event
// record / wait pairs that fork and join the second queue cost ~6.5 us each on this stack, twice per layer on the
// critical path.)  Blocks come in groups of 16: the first 8 run the data gradient, the other 8 the weight gradient, so
// that pair j of either role lands on the same XCD (block b runs on XCD b % 8) and shares its L2 copy of what both stage.
template <int SD, int SW, int HALO = 0>
__global__ __launch_bounds__(CONV_THREADS) void bwd_pair_kernel(
    StageCfg zc, BwdFold bfold, const float* __restrict__ wp, float* __restrict__ dx, const float* __restrict__ xs,
    const float* __restrict__ xs_stats, float* __restrict__ spart, const float* __restrict__ s_prev,
    const float* __restrict__ in_stats, float* __restrict__ wpart, int B, int H, int nblk, WFold wf, StripGeom sg) {
    // groups of 8 * (SD + SW) blocks: utterance group j = 8 * (group index) + x on XCD x gets SD data-gradient workgroups
    // (position slices) and SW weight-gradient workgroups (N-tile slices); SD = SW = 1 at full batches
    const int x = blockIdx.x & 7, y = blockIdx.x >> 3;
    const int r = y % (SD + SW);
    const int j = (y / (SD + SW)) * 8 + x;
    if (j >= nblk) return;
    if (r < SD)
        conv3x3_body<1, SD, HALO>(zc, nullptr, wp, nullptr, dx, xs, xs_stats, spart, nullptr, B, H, j, nblk, r, BnFold{}, bfold, wf, sg);
    else
        wgrad_body<SW, HALO>(WStage{zc, s_prev, false}, in_stats, bfold, wpart, B, H, j, nblk, r - SD, sg);
}

// Deterministic sum over the per-workgroup partial rows: part[g][col], g < nparts.  A block owns 64 columns
// (coalesced reads); its 4 waves split the rows, then combine through LDS in a fixed order.
//   mode 0: out[col] = sum            (conv0: 405 columns)
//   mode 1: col = (cout, tap, cin) of the wgrad accumulator layout [48][9][48] -> dW[(cout*45 + cin)*9 + tap]
// Optional optimiser step in the backward's LAST launch (round 5, single replica): the rows this launch folds get their AdamW
// update from the sum just written (`out - g0 + index` = the element's place in the flat parameter / moment buffers), and one
// more row of blocks walks the rest of the flat buffer [rest_lo, rest_hi) -- the gradients that were final before this launch.
struct RowsAdamW {
    float* p;
    const float* g0;
    float* m;
    float* v;
    HowlAdamWCoef c;
    long rest_lo, rest_hi;
    int on;
};
__device__ __forceinline__ void reduce_rows_body(const float* __restrict__ part, int nparts, int ncols, int mode,
                                                 float* __restrict__ out, const RowsAdamW& opt) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6, nrg = blockDim.x >> 6;     // up to 16 row groups
    const int col = blockIdx.x * 64 + lane;
    float s = 0.0f;
    if (col < ncols) {
        const float* src = part + col;
        // eight rows in flight per lane (a wave's share of the partial rows is a chain of dependent round trips otherwise);
        // the order of the additions is that of the rows either way
        for (int g = rg; g < nparts; g += 8 * nrg) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = g + u * nrg;
                v[u] = src[(size_t)(r < nparts ? r : nparts - 1) * ncols];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (g + u * nrg < nparts) ? v[u] : 0.0f;
        }
    }
    red[rg][lane] = s;
    __syncthreads();
    if (rg == 0 && col < ncols) {
        float tot = red[0][lane];
        for (int r = 1; r < nrg; ++r) tot += red[r][lane];
        long d = -1;
        if (mode == 0) {
            d = col;
        } else {
            const int co = col / WNCOL, idx = col - co * WNCOL;      // idx = 45 tap + cin
            if (co < NMAP && idx < WTAPS) d = (co * NMAP + idx % NMAP) * 9 + idx / NMAP;
        }
        if (d >= 0) {
            out[d] = tot;
            if (opt.on) howl_adamw_element(opt.p, opt.m, opt.v, (size_t)((out - opt.g0) + d), tot, opt.c);
        }
    }
}

// every weight gradient of the backward pass in one launch: blockIdx.y = 0..5 -> conv layer y+1 (partials of layer l at
// part + l * layer_stride), blockIdx.y = 6 -> conv0 (its own partial rows)
__global__ __launch_bounds__(1024) void reduce_rows_all_kernel(const float* __restrict__ part, size_t layer_stride, int nparts,
                                                              HowlPtrs6 out, const float* __restrict__ c0part, int c0parts,
                                                              float* __restrict__ c0out, int y0, int yskip, int nrows, RowsAdamW opt) {
    // a launch covers rows y0, y0 + 1 + yskip, ... of {layer 1..6, conv0} (layers 2..6 are folded inside the pair launches);
    // blockIdx.y == nrows (one row more, with the optimiser step): the parameters whose gradients were final already
    if ((int)blockIdx.y >= nrows) {
        for (long i = opt.rest_lo + (long)blockIdx.x * blockDim.x + threadIdx.x; i < opt.rest_hi; i += (long)gridDim.x * blockDim.x)
            howl_adamw_element(opt.p, opt.m, opt.v, (size_t)i, opt.g0[i], opt.c);
        return;
    }
    const int y = y0 + (int)blockIdx.y * (1 + yskip);
    if (y < 6) {
        reduce_rows_body(part + y * layer_stride, nparts, CP * WNCOL, 1, out.p[y], opt);
    } else if (blockIdx.x * 64 < NMAP * 9) {
        reduce_rows_body(c0part, c0parts, NMAP * 9, 0, c0out, opt);
    }
}

// ---------------------------------------------------------------------------------------------------------
// BatchNorm statistics
// ---------------------------------------------------------------------------------------------------------
// eval mode: stats from the running buffers
__global__ void bn_eval_stats_kernel(HowlPtrs6 rmean, HowlPtrs6 rvar, float* __restrict__ stats) {
    const int layer = blockIdx.x, c = threadIdx.x;
    if (c >= CP) return;
    float* st = stats + (size_t)layer * 2 * CP;
    st[c] = (c < NMAP) ? rmean.p[layer][c] : 0.0f;
    st[CP + c] = (c < NMAP) ? 1.0f / sqrtf(rvar.p[layer][c] + BN_EPS) : 0.0f;
}

// backward elementwise: BatchNorm backward (batch statistics) + skip gradient + ReLU mask
//   ds = rstd * (dx - m1 - xhat * m2) [+ dskip];  dz = ds * mask,  mask = (s > 0) for odd layers and the sign bit of
//   the stored s for the layers with a residual add (see conv_utterance); xhat = (|s| - mean) * rstd
// Two 1024-thread workgroups per CU (a bandwidth-bound sweep wants every wave slot).  The two means m1 = sum dx / N, m2 = sum dx*xhat / N come either ready-made (`m12`, layer
// 6: from the head) or as the data-gradient kernel's per-workgroup partials `part` [nparts][2][48], which every workgroup
// folds itself before its sweep (wave w: channels 3w..3w+2; fold_part_column -- the same bits in every workgroup; 49 KB of coalesced L2
// reads per workgroup instead of a one-block kernel between two launches).
constexpr int BRB_THREADS = 1024;
__global__ __launch_bounds__(BRB_THREADS) void bn_relu_bwd_kernel(
    const float* __restrict__ dx,      // (B,45,P) or nullptr -> broadcast of dpool
    const float* __restrict__ dpool,   // (B,48) used when dx == nullptr, scaled by 1/P
    const float* __restrict__ s, const float* __restrict__ stats, const float* __restrict__ m12,
    const float* __restrict__ part, int nparts, double count,
    const float* __restrict__ dskip,   // nullable
    int even,                          // layer has a residual add: mask in the sign bit of s
    float* __restrict__ ds_out,        // nullable
    float* __restrict__ dz_out, int B, int P) {
    __shared__ float lm[4 * CP];       // mean, rstd, m1, m2
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < CP) {
        lm[tid] = stats[tid];
        lm[CP + tid] = stats[CP + tid];
        if (part == nullptr) {
            lm[2 * CP + tid] = m12[tid];
            lm[3 * CP + tid] = m12[CP + tid];
        }
    }
    if (part != nullptr) {
        const int c8 = lane >> 3;
        const int ch = 3 * wave + (c8 & 3);
        const bool used = (c8 & 3) < 3;
        const double acc = fold_part_column(part, part_stride(nparts), nparts, (c8 < 4 ? 0 : CP) + (used ? ch : 0), lane);
        const double second = __shfl_xor(acc, 32);     // column groups 0..2: sum dx, their partners 4..6: sum dx*xhat
        if (c8 < 3 && (lane & 7) == 0) {
            lm[2 * CP + ch] = (float)(acc / count);
            lm[3 * CP + ch] = (float)(second / count);
        }
    }
    __syncthreads();
    const unsigned n2 = (unsigned)((size_t)B * NMAP * P / 2);
    const float invP = 1.0f / (float)P;
    const unsigned stride = gridDim.x * BRB_THREADS;
    // one element pair: everything after the loads
    auto finish = [&](float2 g, float2 sv, float2 kk, unsigned c, float2& ds, float2& dz) {
        const float mean = lm[c], rstd = lm[CP + c], m1 = lm[2 * CP + c], m2 = lm[3 * CP + c];
        ds.x = rstd * (g.x - m1 - ((fabsf(sv.x) - mean) * rstd) * m2) + kk.x;
        ds.y = rstd * (g.y - m1 - ((fabsf(sv.y) - mean) * rstd) * m2) + kk.y;
        const bool k0 = even ? (sv.x < 0.0f) : (sv.x > 0.0f), k1 = even ? (sv.y < 0.0f) : (sv.y > 0.0f);
        dz = make_float2(k0 ? ds.x : 0.0f, k1 ? ds.y : 0.0f);
    };
    // 16-byte accesses: a thread takes two consecutive element pairs (P is even, so a pair never straddles a channel; the two
    // pairs of a quad may), two quads per trip with all loads requested before the arithmetic (the second index is clamped,
    // its stores masked).  (Measured against the 8-byte version of round 2 on one box: the same 16-25 us per launch, 4.7-5 TB/s --
    // the sweep is not limited by the access width.)
    const unsigned n4 = n2 / 2;
    for (unsigned i0 = blockIdx.x * BRB_THREADS + tid; i0 < n4; i0 += 2 * stride) {
        unsigned idx[2];
        bool ok[2];
        float4 g[2], sv[2], kk[2];
        unsigned cc[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const unsigned i = i0 + u * stride;
            ok[u] = i < n4;
            idx[u] = ok[u] ? i : i0;
            float gb[2] = {0.0f, 0.0f};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned bc = (4u * idx[u] + 2u * h) / (unsigned)P;
                const unsigned bb = bc / NMAP;
                cc[u][h] = bc - bb * NMAP;
                if (dx == nullptr) gb[h] = dpool[bb * CP + cc[u][h]] * invP;
            }
            g[u] = dx != nullptr ? reinterpret_cast<const float4*>(dx)[idx[u]] : make_float4(gb[0], gb[0], gb[1], gb[1]);
            sv[u] = reinterpret_cast<const float4*>(s)[idx[u]];
            kk[u] = dskip != nullptr ? reinterpret_cast<const float4*>(dskip)[idx[u]] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float2 dsa, dza, dsb, dzb;
            finish(make_float2(g[u].x, g[u].y), make_float2(sv[u].x, sv[u].y), make_float2(kk[u].x, kk[u].y), cc[u][0], dsa, dza);
            finish(make_float2(g[u].z, g[u].w), make_float2(sv[u].z, sv[u].w), make_float2(kk[u].z, kk[u].w), cc[u][1], dsb, dzb);
            if (ok[u]) {
                if (ds_out != nullptr) reinterpret_cast<float4*>(ds_out)[idx[u]] = make_float4(dsa.x, dsa.y, dsb.x, dsb.y);
                reinterpret_cast<float4*>(dz_out)[idx[u]] = make_float4(dza.x, dza.y, dzb.x, dzb.y);
            }
        }
    }
    // an odd number of pairs (odd batch): the last pair on its own
    if ((n2 & 1u) && blockIdx.x == 0 && tid == 0) {
        const unsigned i = n2 - 1;
        const unsigned bc = (2u * i) / (unsigned)P;
        const unsigned bb = bc / NMAP, c = bc - bb * NMAP;
        float2 g;
        if (dx != nullptr) {
            g = reinterpret_cast<const float2*>(dx)[i];
        } else {
            const float v = dpool[bb * CP + c] * invP;
            g = make_float2(v, v);
        }
        const float2 sv = reinterpret_cast<const float2*>(s)[i];
        const float2 kk = dskip != nullptr ? reinterpret_cast<const float2*>(dskip)[i] : make_float2(0.0f, 0.0f);
        float2 ds, dz;
        finish(g, sv, kk, c, ds, dz);
        if (ds_out != nullptr) reinterpret_cast<float2*>(ds_out)[i] = ds;
        reinterpret_cast<float2*>(dz_out)[i] = dz;
    }
}

// ---------------------------------------------------------------------------------------------------------
// conv0 (1 -> 45, 3x3, pad 1) + ReLU + AvgPool(3,4), forward and weight gradient: 2.6 MFLOP/utterance each.
// ---------------------------------------------------------------------------------------------------------

// tin[(T+2)][M+4] with a zero halo; the row pitch is a multiple of 4 floats so that the 6-wide patch row of pooled
// column pw (tile columns 4pw .. 4pw+5) is one aligned ds_read_b128 + one ds_read_b64 instead of six strided b32 reads
// (t_lo, t_hi: frames of the tile that exist; a plain utterance or window: 0 .. T, everything else is the convolution's zero
// padding.  A ROW STRIP of a longer clip (StripGeom) also sees the real frames next to it: -1 .. T + 1 clipped to the clip)
template <bool EXACT = false>
__device__ __forceinline__ void load_feat_tile(float* tin, const float* feat, long sb, long st, long sm, int b, int T,
                                               int M, int tid, int nthreads, int t_lo = 0, int t_hi = 0) {
    const int pitch = M + 4;
    const int n = (T + 2) * pitch;
    // batches of 8 independent loads per thread: a one-load-per-iteration loop is bound by HBM latency, not bandwidth
    for (int i0 = tid; i0 < n; i0 += 8 * nthreads) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = i0 + j * nthreads;
            const int t = i / pitch - 1, m = i % pitch - 1;
            const bool ok = i < n && (EXACT ? t >= t_lo && t < t_hi : t >= 0 && t < T) && m >= 0 && m < M;
            // clamped unconditional loads (halo / tail slots read element (b,0,0)), zeroed below: predicated loads were
            // compiled into a chain with vmcnt(0) waits between them, i.e. several HBM round trips back to back
            v[j] = feat[b * sb + (ok ? t * st + m * sm : 0)];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = i0 + j * nthreads;
            const int t = i / pitch - 1, m = i % pitch - 1;
            const bool ok = (EXACT ? t >= t_lo && t < t_hi : t >= 0 && t < T) && m >= 0 && m < M;
            if (i < n) tin[i] = ok ? v[j] : 0.0f;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// The forward runs on the matrix cores: D[position][cout] = sum_tap patch[position][tap] * w[tap][cout] (K = 9 taps padded to
// 12: 3 k-steps; ~2,200 MFMAs per utterance = ~16 us of matrix-pipe time per launch at B = 512 x 1 s).  The ReLU pattern of the
// pre-pool activation goes to the backward pass as 12-bit masks (2 B per pooled output instead of the 583 KB/utterance tensor or
// a recomputation).  The weight gradient runs on the vector pipe (conv0_wgrad_valu_kernel below).
// ---------------------------------------------------------------------------------------------------------
constexpr int C0M_THREADS = 512;   // 8 waves, two per SIMD

// Forward tile = one frame of FOUR neighbouring pooled cells: row i = 4*pwl + fl <-> mel bin 16*blk + i of frame
// 3ph + tl.  The three frames of a cell are three tiles accumulated by the SAME lanes, and a lane holds the 4 mel bins of
// one cell (rows 4g + r, g = pwl): ReLU, the 3x4 sum and the 12 mask bits are all lane-local, no cross-lane traffic.
// 40 mel bins = 2.5 blocks of 16: the third block computes two cells of padding.
// Workgroups beyond the first `nconv` do an unrelated, independent job in the same launch: they rebuild the packed 3x3
// weight fragments of the six following layers (the weights may have changed since the last call), which the first 3x3
// convolution needs only after this kernel has finished anyway.
// NS = strips of 10 pooled columns per utterance (1: 40 mel bins; 2: 80, written as the two blocks v = 2 b + strip, see HaloSlot)
// EXACT: the nwin "windows" of a clip are ROW STRIPS of a training utterance (StripGeom): window wi starts at frame wi * win_step,
// sees the clip's real frames on both sides (win_last = the clip's frame count) and is block (clip * nwin + wi) of the output
template <int NS = 1, bool EXACT = false>
__global__ __launch_bounds__(C0M_THREADS) void conv0_fwd_mfma_kernel(const float* __restrict__ feat, long sb, long st,
                                                                    long sm, const float* __restrict__ w0,
                                                                    float* __restrict__ s0, unsigned short* __restrict__ mask0,
                                                                    int B, int T, int M, int H, int nconv, HowlPtrs6 cw,
                                                                    float* __restrict__ wp_fwd, float* __restrict__ wp_bwd,
                                                                    int nwin, int win_step, int win_last, int slices) {
    if ((int)blockIdx.x >= nconv) {
        const int e = ((int)blockIdx.x - nconv) * C0M_THREADS + (int)threadIdx.x;   // (mode, layer, fragment element)
        if (e < 2 * 6 * PACK_ELEMS) pack_weights_one(cw, wp_fwd, wp_bwd, (e / PACK_ELEMS) % 6, e / (6 * PACK_ELEMS), e % PACK_ELEMS);
        return;
    }
    HIP_DYNAMIC_SHARED(float, lds)
    float* tin = lds;  // (T+2) x (M+4), zero halo (+ conv0_tile_floats' slack for the second row of an odd last pair)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pitch = M + 4;
    const int P = H * PW;
    const int g = lane >> 4, n = lane & 15;
    // B fragments: B[k = tap = 4ks + g][col = cout = 16nt + n]
    float bw[3][3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int tap = 4 * ks + g, c = 16 * nt + n;
            bw[ks][nt] = (tap < 9 && c < NMAP) ? w0[c * 9 + tap] : 0.0f;
        }
    // A fragments: a 16-row tile is 8 mel bins (two neighbouring pooled cells) of one frame of TWO consecutive pooled rows --
    // 40 mel bins are exactly five groups of eight (rounds 2-3 used 16 bins of one pooled row: 2.5 blocks, the third half
    // padding: 81 units per utterance, now 70) -- row i = n: pooled row +(n >> 3), mel bin 8j + (n & 7):
    //   A[i][k = tap = 4ks + g] = tin[frame 3(ph + (n >> 3)) + tl + tap/3][mel 8j + (n & 7) + tap%3]  (halo origin)
    int aoff[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        const int tap = min(4 * ks + g, 8);  // taps 9..11 meet zero weights: any finite value will do
        aoff[ks] = (tap / 3) * pitch + tap % 3 + (n & 7) + (n >> 3) * 3 * pitch;
    }
    for (int i = tid; i < 3 * pitch + 16; i += C0M_THREADS) tin[(T + 2) * pitch + i] = 0.0f;  // slack read by an odd last row pair
    // Small batches: `slices` workgroups share an utterance's pooled rows (one utterance costs a workgroup ~20 us whatever the
    // batch: at B <= 64 that was the whole launch with three quarters of the CUs idle); work item = (utterance, slice).
    for (int item = blockIdx.x; item < B * slices; item += nconv) {
        const int b = item / slices, sl = item - b * slices;
        const int ph0 = (sl * H) / slices, ph1 = ((sl + 1) * H) / slices;
        __syncthreads();
        // long inputs (howl_res8_fwd_long): "utterance" b is window b % nwin of clip b / nwin, T frames from its start frame
        const int clip = b / nwin, wi = b - clip * nwin;
        const int t0 = EXACT ? wi * win_step : min(wi * win_step, win_last);
        load_feat_tile<EXACT>(tin, feat + (long)clip * sb + (long)t0 * st, 0, st, sm, 0, T, M, tid, C0M_THREADS, -t0, win_last - t0);
        __syncthreads();
        // units (pair of pooled rows, group of 8 mel bins) of this slice, dealt to the waves
        constexpr int NJ = 5 * NS;      // groups of eight mel bins
        for (int u = wave; u < NJ * ((ph1 - ph0 + 1) / 2); u += C0M_THREADS / 64) {
            const int pp = u / NJ, j8 = u - NJ * pp;
            const float* rowp = tin + 3 * (ph0 + 2 * pp) * pitch + 8 * j8;
            {
                f32x4 acc[3][3];  // [frame tl][cout tile nt]
#pragma unroll
                for (int tl = 0; tl < 3; ++tl)
#pragma unroll
                    for (int nt = 0; nt < 3; ++nt) acc[tl][nt] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int tl = 0; tl < 3; ++tl)
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks) {
                        const float a = rowp[aoff[ks] + tl * pitch];
#pragma unroll
                        for (int nt = 0; nt < 3; ++nt)
                            acc[tl][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[ks][nt], acc[tl][nt], 0, 0, 0);
                    }
                const int pw = 2 * j8 + (g & 1), ph = ph0 + 2 * pp + (g >> 1);  // this lane's cell: rows 4g..4g+3 of the tile
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) {
                    float sum = 0.0f;
                    unsigned bits = 0;  // bit (4 tl + fl): pre-pool activation (3ph + tl, 4pw + fl) is positive
#pragma unroll
                    for (int tl = 0; tl < 3; ++tl)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            // relu(z) as an INTEGER max (negatives and -0 -> +0, exact) and the mask bit as min(bits, 1):
                            // four vector instructions per value (the compare / select / shift / or form took six, and
                            // this epilogue does not overlap the MFMAs: 18 of the kernel's 43 us)
                            const int rb = max(__float_as_int(acc[tl][nt][r]), 0);
                            sum += __int_as_float(rb);
                            bits |= min((unsigned)rb, 1u) << (4 * tl + r);
                        }
                    const int c = 16 * nt + n;
                    if (ph < ph1 && c < NMAP) {
                        const int strip = NS > 1 ? pw / PW : 0;
                        const size_t o = (((size_t)b * NS + strip) * NMAP + c) * P + (size_t)ph * PW + (pw - strip * PW);
                        s0[o] = sum * (1.0f / 12.0f);
                        if (mask0 != nullptr) mask0[o] = (unsigned short)bits;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// conv0's weight gradient on the vector pipe (round 4).  The matrix-core version of rounds 2-3 (GEMM cout x taps over
// positions: 9 useful tap columns of 16, the A operand rebuilt from mask bits per cell) took 50.6 us per launch at 512 x 1 s;
// this one 48.4 us.  lane = pooled cell, wave = three output channels:
//   * a lane keeps its cell's 5 x 6 input patch in registers (five aligned 16 + 8 byte LDS reads) and the 9 accumulators of each
//     of its wave's channels across every cell block and utterance of the workgroup: 108 FMAs per (cell, channel), no padding;
//   * consecutive lanes are consecutive cells = consecutive addresses of dx_0 / ds_2 / mask0: every global access is a
//     coalesced row, straight from HBM (no LDS staging of the gradients);
//   * the forward's ReLU pattern is applied as bfe + and (mask bit -> 0 / -1 -> gradient or +0).
// The same formulation of the FORWARD (lane = cell, weights in scalar registers, coalesced stores) was built and measured at
// 49.5 us against 43.3 us for conv0_fwd_mfma_kernel -- the vector pipe needs ~4 cycles per instruction here, not the 2.2-2.5 of
// the microbenchmark -- and removed (tools/variants4.py: MFMA part 16.4 us, per-value epilogue 18 us, scattered stores 9 us,
// and the three ADD UP: vector and matrix work of different waves do not overlap on this part).
// ---------------------------------------------------------------------------------------------------------
// weight gradient: its accumulators (9 per channel) stay in registers, so 3 channels per wave on 15 waves: one workgroup per CU
// puts 4 / 4 / 4 / 3 waves on the SIMDs (nine 5-channel waves would be 3 / 2 / 2 / 2 at 168 VGPRs)
constexpr int C0G_WAVES = 15, C0G_THREADS = 64 * C0G_WAVES, C0G_CPW = NMAP / C0G_WAVES;
static_assert(C0G_CPW * C0G_WAVES == NMAP, "channels split evenly over the waves");

__device__ __forceinline__ void load_patch(float (&x)[5][6], const float* tin, int pitch, int ph, int pw) {
    const float* base = tin + 3 * ph * pitch + 4 * pw;      // frames 3ph-1 .. 3ph+3, mel bins 4pw-1 .. 4pw+4 (halo origin)
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const float4 a = *reinterpret_cast<const float4*>(base + r * pitch);
        const float2 c = *reinterpret_cast<const float2*>(base + r * pitch + 4);
        x[r][0] = a.x, x[r][1] = a.y, x[r][2] = a.z, x[r][3] = a.w, x[r][4] = c.x, x[r][5] = c.y;
    }
}

// weight gradient: dW0[c][tap] = sum over utterances, cells, the 12 positions of a cell of  (g[c][cell] / 12 where the forward's
// mask bit is set) * x[position + tap];  acc[5][9] per lane (its wave's channels) lives in registers across every cell block and
// utterance of the workgroup, one wave-wide sum per accumulator at the end -> one partial row per workgroup.
// EXACT (StripGeom): every utterance is sg.nr row strips of H pooled rows (T = 3 H frames each, the clip has t_total), the last
// one with sg.hv_last valid rows
template <int NS = 1, bool EXACT = false>
__global__ __launch_bounds__(C0G_THREADS) void conv0_wgrad_valu_kernel(const float* __restrict__ feat, long sb, long st, long sm,
                                                                      const unsigned short* __restrict__ mask0,
                                                                      const float* __restrict__ ga, const float* __restrict__ gb,
                                                                      float* __restrict__ part, int B, int T, int M, int H,
                                                                      int slices, StripGeom sg, int t_total) {
    HIP_DYNAMIC_SHARED(float, lds)
    float* tin = lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pitch = M + 4;
    const int P = H * PW;
    float acc[C0G_CPW][9];
#pragma unroll
    for (int j = 0; j < C0G_CPW; ++j)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[j][t] = 0.0f;
    // work item = (utterance, strip of 10 pooled columns, slice of the strip's cells); B counts utterances
    const int nrs = EXACT ? sg.nr : 1;
    for (int item = blockIdx.x; item < B * nrs * NS * slices; item += gridDim.x) {
        const int vb = item / slices, sl = item - vb * slices;   // gradient / mask block vb = (utterance * nr + row strip) * NS + strip
        const int vq = NS > 1 ? vb / NS : vb, strip = NS > 1 ? vb - vq * NS : 0;
        const int b = EXACT ? vq / nrs : vq, rs = EXACT ? vq - b * nrs : 0;
        const int Pv = (EXACT && rs == nrs - 1) ? sg.hv_last * PW : P;      // cells of the strip that lie inside the map
        const int cell0 = (sl * Pv) / slices, cell1 = ((sl + 1) * Pv) / slices;
        __syncthreads();
        if constexpr (EXACT)
            load_feat_tile<true>(tin, feat + (long)rs * T * st, sb, st, sm, b, T, M, tid, C0G_THREADS, -rs * T, t_total - rs * T);
        else
            load_feat_tile(tin, feat, sb, st, sm, b, T, M, tid, C0G_THREADS);
        __syncthreads();
        const float* gab = ga + ((size_t)vb * NMAP + C0G_CPW * wave) * P;      // (uniform bases, 32-bit lane offsets)
        const float* gbb = gb != nullptr ? gb + ((size_t)vb * NMAP + C0G_CPW * wave) * P : nullptr;
        const unsigned short* mb = mask0 + ((size_t)vb * NMAP + C0G_CPW * wave) * P;
        for (int cb = cell0; cb < cell1; cb += 64) {
            const int cell = cb + lane;
            const bool valid = cell < cell1;
            const unsigned cc = (unsigned)(valid ? cell : cell1 - 1);
            float gv[C0G_CPW];
            unsigned mv[C0G_CPW];
#pragma unroll
            // (requesting the NEXT block's operands here instead -- one block ahead -- needs six more registers at the 128-VGPR line of
            // fifteen waves: 8-11 dwords of scratch whose reloads share the in-order memory counter with the prefetch; 60-64 us against 44.8)
            for (int j = 0; j < C0G_CPW; ++j) {      // this block's operands of all five channels in flight together
                const unsigned o = (unsigned)(j * P) + cc;
                gv[j] = gab[o] + (gbb != nullptr ? gbb[o] : 0.0f);
                mv[j] = mb[o];
            }
            const int ph = (int)cc / PW, pw = (int)cc - ph * PW;
            float x[5][6];
            load_patch(x, tin, pitch, ph, pw + strip * PW);
#pragma unroll
            for (int j = 0; j < C0G_CPW; ++j) {
                const int gbits = valid ? __float_as_int(gv[j] * (1.0f / 12.0f)) : 0;
#pragma unroll
                for (int tl = 0; tl < 3; ++tl)
#pragma unroll
                    for (int fl = 0; fl < 4; ++fl) {
                        const int k = 4 * tl + fl;
                        const int sel = ((int)(mv[j] << (31 - k))) >> 31;           // 0 or -1: the forward's ReLU pattern
                        const float dy = __int_as_float(sel & gbits);
#pragma unroll
                        for (int tap = 0; tap < 9; ++tap) acc[j][tap] = fmaf(dy, x[tl + tap / 3][fl + tap % 3], acc[j][tap]);
                    }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < C0G_CPW; ++j)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float v = wave_sum_rows(acc[j][t]);
            if (lane == 0) part[(size_t)blockIdx.x * NMAP * 9 + (C0G_CPW * wave + j) * 9 + t] = v;
        }
}

// ---------------------------------------------------------------------------------------------------------
// head: BN6 -> spatial mean -> Linear(45, C)   (cnn.py:143-145), and its backward
// ---------------------------------------------------------------------------------------------------------
// With `labels` the same launch is also the loss of the training step, utterance by utterance (what howl_xent_fwd_bwd and the
// first launch of the backward pass would do, same arithmetic in the same order): nll[b] = lse - logit[label],
// dlogits[b] = (softmax - onehot) * inv_batch, dpool[b] = dlogits[b] . W_out.  The batch mean of nll is taken by
// head_bwd_param_kernel.  Needs C <= HEAD_XC classes.
constexpr int HEAD_XC = 64;
// Round 6: twelve waves (every wave folds ONE quad of statistics columns: the three dependent fold trips of a four-wave block were
// 5 us of a 15-us launch at batch 64), the output layer's weights in LDS (the logits and the pooled gradient were chains of global
// loads), the per-group pooled sums fetched by all threads at once and added in the same order as before (bit-identical results).
constexpr int HEAD_THREADS = 768, HEAD_PARTS = 16;
__global__ __launch_bounds__(HEAD_THREADS) void head_fwd_kernel(const float* __restrict__ s6, const float* __restrict__ stats,
                                                       const float* __restrict__ wout, const float* __restrict__ bout,
                                                       float* __restrict__ pooled, float* __restrict__ logits, int B,
                                                       int P, int C, const long long* __restrict__ labels,
                                                       float* __restrict__ nll, float* __restrict__ dlogits,
                                                       float* __restrict__ dpool, float inv_batch,
                                                       const float* __restrict__ pool, int npg, int npg_used, BnFold fold,
                                                       int ns /* strips per utterance: pool is [B * ns][npg][48], P counts all of them */) {
    __shared__ float lp[CP];
    __shared__ float ll[HEAD_XC], dl[HEAD_XC], lse_s;
    __shared__ float lst[2 * CP];
    __shared__ float lw[HEAD_XC * NMAP];          // output.weight when C <= HEAD_XC
    __shared__ float lparts[HEAD_PARTS][CP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool w_in_lds = C <= HEAD_XC;
    if (w_in_lds)
        for (int i = tid; i < C * NMAP; i += HEAD_THREADS) lw[i] = wout[i];
    if (fold.part != nullptr) {
        // training: BatchNorm 6's batch statistics from the last convolution's partials, folded by every workgroup (the same bits
        // everywhere: fold_part_column) -- without a one-block finalize launch between the last
        // convolution and the head; workgroup 0 publishes them for the backward pass and updates the running buffers (cnn.py:142)
        const int c8 = lane >> 3;
        for (int w0 = wave; w0 < CP / 4; w0 += HEAD_THREADS / 64) {
            const int c = 4 * w0 + (c8 & 3);
            const double sm = fold_part_column(fold.part, part_stride(fold.nparts), fold.nparts, (c8 < 4 ? 0 : CP) + c, lane);
            const double q = __shfl_xor(sm, 32);
            if (c8 < 4 && (lane & 7) == 0) {
                const double mean = sm / fold.count;
                double var = q / fold.count - mean * mean;
                var = var < 0.0 ? 0.0 : var;
                const float fm = (c < NMAP) ? (float)mean : 0.0f;
                const float fr = (c < NMAP) ? (float)(1.0 / sqrt(var + (double)BN_EPS)) : 0.0f;
                lst[c] = fm;
                lst[CP + c] = fr;
                if (blockIdx.x == 0) {
                    fold.stats_out[c] = fm;
                    fold.stats_out[CP + c] = fr;
                    if (c < NMAP && fold.bn.running_mean != nullptr) {
                        const double unbiased = fold.count > 1.0 ? var * fold.count / (fold.count - 1.0) : var;
                        fold.bn.running_mean[c] = (1.0f - BN_MOMENTUM) * fold.bn.running_mean[c] + BN_MOMENTUM * (float)mean;
                        fold.bn.running_var[c] = (1.0f - BN_MOMENTUM) * fold.bn.running_var[c] + BN_MOMENTUM * (float)unbiased;
                    }
                    if (c == 0 && fold.bn.num_batches != nullptr) fold.bn.num_batches[0] += 1;
                }
            }
        }
    } else if (tid < 2 * CP) {
        lst[tid] = stats[tid];
    }
    __syncthreads();
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
        if (pool != nullptr) {
            // the last convolution left the sums of |s_6| per (utterance, position group, channel): all of an utterance's parts are
            // requested at once (thread = (part, channel)), then added in part order by the channel's thread
            const int nparts = ns * npg_used;
            const int pc = tid % CP, pk = tid / CP;      // 16 parts x 48 channels = 768 threads
            float acc = 0.0f;
            for (int k0 = 0; k0 < nparts; k0 += HEAD_PARTS) {
                const int k = k0 + pk;
                if (k < nparts) {
                    const int sgi = k / npg_used, g = k - sgi * npg_used;
                    lparts[pk][pc] = pool[(((size_t)b * ns + sgi) * npg + g) * CP + pc];
                }
                __syncthreads();
                if (tid < CP) {
                    const int n = nparts - k0 < HEAD_PARTS ? nparts - k0 : HEAD_PARTS;
                    for (int j = 0; j < n; ++j) acc += lparts[j][tid];
                }
                __syncthreads();
            }
            if (tid < CP) {
                const float v = tid < NMAP ? (acc / (float)P - lst[tid]) * lst[CP + tid] : 0.0f;
                lp[tid] = v;
                pooled[(size_t)b * CP + tid] = v;
            }
        } else if (wave < 4)
        // a wave owns channels wave, wave+4, ...: three of them per trip (24 loads in flight) -- one channel per trip is a
        // chain of twelve load latencies
        for (int c0 = wave; c0 < CP; c0 += 12) {
            float a8[3][8];   // <= 8 x 64 positions per channel (P <= 270); clamped addresses instead of predicated loads,
                              // which would each wait for their own data
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int c = c0 + 4 * u < NMAP ? c0 + 4 * u : NMAP - 1;
                const float* src = s6 + ((size_t)b * NMAP + c) * P;
#pragma unroll
                for (int j = 0; j < 8; ++j) a8[u][j] = src[min(lane + 64 * j, P - 1)];
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int c = c0 + 4 * u;
#pragma unroll
                for (int j = 0; j < 8; ++j) a8[u][j] = (lane + 64 * j < P) ? fabsf(a8[u][j]) : 0.0f;  // |s|: see conv_utterance
                float acc = ((a8[u][0] + a8[u][1]) + (a8[u][2] + a8[u][3])) + ((a8[u][4] + a8[u][5]) + (a8[u][6] + a8[u][7]));
                acc = wave_sum(acc);
                const float v = c < NMAP ? (acc / (float)P - lst[c]) * lst[CP + c] : 0.0f;
                if (lane == 0 && c < CP) {
                    lp[c] = v;
                    pooled[(size_t)b * CP + c] = v;
                }
            }
        }
        __syncthreads();
        for (int k = tid; k < C; k += HEAD_THREADS) {
            float acc = bout[k];
            if (w_in_lds)
                for (int c = 0; c < NMAP; ++c) acc = fmaf(lw[k * NMAP + c], lp[c], acc);
            else
                for (int c = 0; c < NMAP; ++c) acc = fmaf(wout[k * NMAP + c], lp[c], acc);
            logits[(size_t)b * C + k] = acc;
            if (labels != nullptr) ll[k] = acc;
        }
        if (labels == nullptr) continue;       // (uniform)
        __syncthreads();
        const int y = (int)labels[b];
        if (tid == 0) {                        // the row's log-sum-exp exactly as xent_kernel takes it (serial, in class order)
            float mx = ll[0];
            for (int k = 1; k < C; ++k) mx = fmaxf(mx, ll[k]);
            float se = 0.0f;
            for (int k = 0; k < C; ++k) se += expf(ll[k] - mx);
            const float lse = mx + logf(se);
            lse_s = lse;
            nll[b] = lse - ll[y];
        }
        __syncthreads();
        if (tid < C) {
            const float d = (expf(ll[tid] - lse_s) - (tid == y ? 1.0f : 0.0f)) * inv_batch;
            dl[tid] = d;
            dlogits[(size_t)b * C + tid] = d;
        }
        __syncthreads();
        if (tid < CP) {                        // head_bwd_pool_kernel's sum (labels given: C <= HEAD_XC, the weights are in LDS)
            float acc = 0.0f;
            if (tid < NMAP)
                for (int k = 0; k < C; ++k) acc = fmaf(dl[k], lw[k * NMAP + tid], acc);
            dpool[(size_t)b * CP + tid] = acc;
        }
    }
}

// Long inputs: clip b was processed as nwin overlapping windows of Hw pooled rows (virtual utterances b*nwin + w); window w
// contributes its rows [lo[w], hi[w]) -- a partition of the clip's rows, away from the windows' own zero padding -- to the
// spatial mean.  Otherwise head_fwd_kernel: BatchNorm (running statistics), mean, Linear.
__global__ __launch_bounds__(256) void head_fwd_windows_kernel(const float* __restrict__ s6, const float* __restrict__ stats,
                                                               const float* __restrict__ wout, const float* __restrict__ bout,
                                                               float* __restrict__ logits, int B, int nwin, int Hw,
                                                               HowlWinRows rows, float inv_count, int C, int ns) {
    __shared__ float lp[CP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Pw = Hw * PW;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
        for (int c = wave; c < CP; c += 4) {
            float acc = 0.0f;
            if (c < NMAP) {
                for (int w = 0; w < nwin * ns; ++w) {      // (window, strip of 10 pooled columns) blocks of clip b
                    const float* src = s6 + ((size_t)(b * nwin * ns + w) * NMAP + c) * Pw;
                    for (int i = rows.lo[w / ns] * PW + lane; i < rows.hi[w / ns] * PW; i += 64) acc += fabsf(src[i]);
                }
            }
            acc = wave_sum(acc);
            if (lane == 0) lp[c] = c < NMAP ? (acc * inv_count - stats[c]) * stats[CP + c] : 0.0f;
        }
        __syncthreads();
        for (int k = tid; k < C; k += 256) {
            float acc = bout[k];
            for (int c = 0; c < NMAP; ++c) acc = fmaf(wout[k * NMAP + c], lp[c], acc);
            logits[(size_t)b * C + k] = acc;
        }
    }
}

// dpool[b][c] = sum_k dlogits[b][k] W[k][c]
__global__ void head_bwd_pool_kernel(const float* __restrict__ dlogits, const float* __restrict__ wout,
                                     float* __restrict__ dpool, int B, int C) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * CP) return;
    const int b = idx / CP, c = idx - b * CP;
    float acc = 0.0f;
    if (c < NMAP)
        for (int k = 0; k < C; ++k) acc = fmaf(dlogits[(size_t)b * C + k], wout[k * NMAP + c], acc);
    dpool[idx] = acc;
}

// one workgroup per output row k (dW_out[k][:], db[k]); block k == C produces the BN6 backward means
// m1[c] = sum_b dpool / N, m2[c] = sum_b dpool*pooled / N  (dx6 is dpool/P broadcast over positions).
// 16 waves split the batch, lanes are channels; fixed-order combine through LDS.
__global__ __launch_bounds__(1024) void head_bwd_param_kernel(const float* __restrict__ dlogits,
                                                              const float* __restrict__ pooled,
                                                              const float* __restrict__ dpool, float* __restrict__ dwout,
                                                              float* __restrict__ dbout, float* __restrict__ m12, int B,
                                                              int C, int P, const float* __restrict__ nll,
                                                              float* __restrict__ loss) {
    __shared__ double red[2][16][64];
    if ((int)blockIdx.x == C + 1) {      // mean of the per-utterance losses, in xent_kernel's summation order
        double acc = 0.0;
        for (int b = threadIdx.x; b < B; b += 1024) acc += (double)nll[b];
        acc = wave_sum_d(acc);
        if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6][0] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot = 0.0;
            for (int w = 0; w < 16; ++w) tot += red[0][w][0];
            loss[0] = (float)(tot / (double)B);
        }
        return;
    }
    const int k = blockIdx.x, c = threadIdx.x & 63, bg = threadIdx.x >> 6;
    double a0 = 0.0, a1 = 0.0;
    // 16 rows per iteration, clamped unconditional loads: a loop with few rows per trip is serialised by the latency of
    // its own loads (B = 512: two trips per wave instead of eight)
    constexpr int HR = 16;
    const int cc = c < CP ? c : CP - 1;
    if (k < C) {
        for (int b = bg; b < B; b += 16 * HR) {
            float d[HR], pv[HR];
#pragma unroll
            for (int j = 0; j < HR; ++j) {
                const int bb = b + 16 * j < B ? b + 16 * j : B - 1;
                d[j] = dlogits[(size_t)bb * C + k];
                pv[j] = pooled[(size_t)bb * CP + cc];
            }
#pragma unroll
            for (int j = 0; j < HR; ++j) {
                const bool ok = b + 16 * j < B;
                a1 += ok ? (double)d[j] : 0.0;
                a0 += (ok && c < CP) ? (double)d[j] * (double)pv[j] : 0.0;
            }
        }
    } else if (c < CP) {
        for (int b = bg; b < B; b += 16 * HR) {
            float d[HR], pv[HR];
#pragma unroll
            for (int j = 0; j < HR; ++j) {
                const int bb = b + 16 * j < B ? b + 16 * j : B - 1;
                d[j] = dpool[(size_t)bb * CP + c];
                pv[j] = pooled[(size_t)bb * CP + c];
            }
#pragma unroll
            for (int j = 0; j < HR; ++j) {
                const bool ok = b + 16 * j < B;
                a0 += ok ? (double)d[j] : 0.0;
                a1 += ok ? (double)d[j] * (double)pv[j] : 0.0;
            }
        }
    }
    red[0][bg][c] = a0;
    red[1][bg][c] = a1;
    __syncthreads();
    if (bg != 0) return;
    double s0 = 0.0, s1 = 0.0;
    for (int g = 0; g < 16; ++g) {
        s0 += red[0][g][c];
        s1 += red[1][g][c];
    }
    if (k < C) {
        if (c < NMAP) dwout[k * NMAP + c] = (float)s0;
        if (c == 0) dbout[k] = (float)s1;
    } else if (c < CP) {
        const double n = (double)B * (double)P;
        m12[c] = (float)(s0 / n);
        m12[CP + c] = (float)(s1 / n);
    }
}

// mean cross-entropy and its gradient (pretrain_gsc.py:95,131; train.py:251,293): one workgroup
__global__ __launch_bounds__(1024) void xent_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                    int B, int C, float* __restrict__ loss, float* __restrict__ dlogits) {
    __shared__ double red[16];
    double acc = 0.0;
    const float invB = 1.0f / (float)B;
    constexpr int XC = 32;      // rows of up to XC classes are held in registers: ONE round trip for the row instead of three
                                // passes of dependent loads (run-time trip counts: ~10 us for 64 x 12 logits)
    for (int b = threadIdx.x; b < B; b += 1024) {
        const float* row = logits + (size_t)b * C;
        const int y = (int)labels[b];
        if (C <= XC) {
            float v[XC];
#pragma unroll
            for (int k = 0; k < XC; ++k) v[k] = row[k < C ? k : C - 1];
            float mx = v[0];
#pragma unroll
            for (int k = 1; k < XC; ++k) mx = k < C ? fmaxf(mx, v[k]) : mx;
            float se = 0.0f;
#pragma unroll
            for (int k = 0; k < XC; ++k) se += k < C ? expf(v[k] - mx) : 0.0f;
            const float lse = mx + logf(se);
            float vy = v[0];
#pragma unroll
            for (int k = 1; k < XC; ++k) vy = k == y ? v[k] : vy;
            acc += (double)(lse - vy);
            if (dlogits != nullptr) {
#pragma unroll
                for (int k = 0; k < XC; ++k)
                    if (k < C) dlogits[(size_t)b * C + k] = (expf(v[k] - lse) - (k == y ? 1.0f : 0.0f)) * invB;
            }
            continue;
        }
        float mx = row[0];
        for (int k = 1; k < C; ++k) mx = fmaxf(mx, row[k]);
        float se = 0.0f;
        for (int k = 0; k < C; ++k) se += expf(row[k] - mx);
        const float lse = mx + logf(se);
        acc += (double)(lse - row[y]);
        if (dlogits != nullptr)
            for (int k = 0; k < C; ++k)
                dlogits[(size_t)b * C + k] = (expf(row[k] - lse) - (k == y ? 1.0f : 0.0f)) * invB;
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < 16; ++w) tot += red[w];
        loss[0] = (float)(tot / (double)B);
    }
}

// fused flat AdamW (torch.optim.AdamW defaults; pretrain_gsc.py:93,133)
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, size_t n, HowlAdamWCoef c) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        howl_adamw_element(p, m, v, i, g[i], c);
}

size_t conv0_tile_floats(int T, int M) { return (size_t)(T + 2) * (M + 4) + 3 * (M + 4) + 16; }   // tile + slack (conv0_fwd_mfma_kernel)
size_t conv_lds_bytes(int H, bool own_rows = false) {
    return (size_t)(3 * KSTEPS * 64 + tile_floats(H, own_rows) + 4 * CP + 12 * 2 * 16) * sizeof(float);
}
StageCfg with_probe(StageCfg c) { return c; }

size_t wgrad_lds_bytes(int H) { return (size_t)(tile_floats_z(H) + tile_floats_x(H) + 6 * CP) * sizeof(float); }

struct Ws {
    float* wp_fwd;   // [6][3][108][64]
    float* wp_bwd;
    float* part;     // statistics partials, transposed [2][48][part_stride(G)] (fold_part_column)
    float* part2;    // second set: a forward layer writes one while the next layer's prologue may still read the other
    float* stats;    // eval-mode stats [6][2][48]
    float* m12;      // [2][48]
    float* dpool;    // [B][48]
    float* pool;     // [B][16][48] per-utterance sums of |s_6| per position group (ConvEpilogue::pool)
    float* bufa;     // (B,45,P) x6: dx ping-pong, dz ping-pong, ds ping-pong
    float* bufb;
    float* dz;
    float* dz2;
    float* dsa;
    float* dsb;
    float* wpart;    // [6][G][48][WNCOL]: one set of weight-gradient partials per layer (reduced together at the end)
    float* c0part;   // [G][405]
};

// `fwd_eval_bytes`: what an eval-mode forward touches (packed weights, statistics, pooled sums: everything in front of the
// backward's activation-sized buffers)
size_t ws_layout(Ws* w, char* base, int B, int H, int G, size_t* fwd_eval_bytes = nullptr) {
    size_t off = 0;
    auto take = [&](size_t floats) {
        float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
        off += ((floats * sizeof(float) + 255) / 256) * 256;
        return p;
    };
    const size_t act = (size_t)B * NMAP * H * PW;
    Ws t;
    t.wp_fwd = take((size_t)6 * 3 * KSTEPS * 64);
    t.wp_bwd = take((size_t)6 * 3 * KSTEPS * 64);
    const int max_parts = G > howl_num_cus() ? G : howl_num_cus();     // one row per workgroup; slicing never exceeds the CU count
    t.part = take((size_t)part_stride(max_parts) * 2 * CP);
    t.part2 = take((size_t)part_stride(max_parts) * 2 * CP);
    t.stats = take((size_t)6 * 2 * CP);
    t.m12 = take(2 * CP);
    t.dpool = take((size_t)B * CP);
    t.pool = take((size_t)B * 16 * CP);
    if (fwd_eval_bytes) *fwd_eval_bytes = off;
    t.bufa = take(act);
    t.bufb = take(act);
    t.dz = take(act);
    t.dz2 = take(act);
    t.dsa = take(act);
    t.dsb = take(act);
    t.wpart = take((size_t)6 * G * CP * WNCOL);
    t.c0part = take((size_t)max_parts * NMAP * 9);     // one row per workgroup of conv0's weight gradient (<= CUs with slicing)
    if (w) *w = t;
    return off;
}

// Backward pass: per layer  bn_relu_bwd -> [dgrad || wgrad in one launch, half the CUs each] -> bn_bwd_finalize, all on the
// caller's stream; every layer's weight-gradient partials stay in the workspace and ONE launch reduces them all at the
// end (nothing but AdamW waits for them).  HOWL_RES8_BWD_PAIR=0 launches the two halves one after the other with the
// same grids (bit-identical results; the reference point of the tests).
int conv_grid(int B) {
    int g = howl_num_cus();
    return B < g ? B : g;
}
// Mel bins -> strips of 10 pooled columns (see HaloSlot): 40 -> 1, 80 -> 2 (the reference's stock NUM_MELS, settings.py:32)
int mel_strips(int M) { return M == 40 ? 1 : (M == 80 ? 2 : 0); }
// a workgroup's utterances b, b + nblk, ... must keep their strip parity on wide maps: an even stride (B = 2 x utterances >= 2)
// How (B, T, M) runs as blocks of (45, Hs, 10): ns column strips (mel_strips) x nr row strips of Hs <= 27 pooled rows, the last row
// strip with hv_last valid rows (StripGeom); halo = 0: plain utterances, 1: column strips only (HaloSlot), 2: row strips
struct Strips {
    int ns, nr, Hs, hv_last, Bv, halo;
    StripGeom sg;
};
Strips strips_for(int B, int T, int M) {
    Strips st;
    const int H = T / 3;
    st.ns = mel_strips(M) > 0 ? mel_strips(M) : 1;
    st.nr = H <= MAX_H ? 1 : (H + MAX_H - 1) / MAX_H;
    st.Hs = (H + st.nr - 1) / st.nr;
    st.hv_last = H - (st.nr - 1) * st.Hs;
    st.Bv = B * st.nr * st.ns;
    st.halo = st.nr > 1 ? 2 : (st.ns > 1 ? 1 : 0);
    st.sg = StripGeom{st.nr, st.ns, st.hv_last};
    return st;
}
int even_grid(int g, int strips) { return strips > 1 ? ((g & ~1) > 2 ? (g & ~1) : 2) : g; }
// Small batches (the reference's presets train at 16, its engines run at batch 1): how many workgroups share one utterance.
// Forward / data gradient split the position tiles (4 or 2 ways: every position group of a workgroup keeps at least one
// tile), the weight gradient its 27 N tiles (2 ways).
int launch_blocks(int nblk, int per_group) { return 8 * per_group * ((nblk + 7) / 8); }
bool slicing_enabled() {     // HOWL_RES8_SLICES=0: one workgroup per utterance whatever the batch (the reference point of the tests)
    const char* e = getenv("HOWL_RES8_SLICES");
    return !(e != nullptr && e[0] == '0');
}
int conv_slices(int nblk, int H, int budget) {
    const int ntiles = (H * PW + 15) / 16;
    if (!slicing_enabled()) return 1;
    for (int sl = 4; sl > 1; sl >>= 1)
        if (nblk * sl <= budget && 4 * sl <= ntiles) return sl;
    return 1;
}
// conv0 (forward and weight gradient): up to eight workgroups share an utterance's pooled rows / cells while B * slices <= CUs
int conv0_slices(int B) {
    if (!slicing_enabled()) return 1;
    int sl = howl_num_cus() / (B > 0 ? B : 1);
    return sl < 1 ? 1 : (sl > 8 ? 8 : sl);
}
void pair_slices(int nblk, int H, int* sd, int* sw) {
    const int cus = howl_num_cus(), ntiles = (H * PW + 15) / 16;
    const int opts[4][2] = {{4, 2}, {2, 2}, {2, 1}, {1, 1}};
    *sd = *sw = 1;
    if (!slicing_enabled()) return;
    for (const auto& o : opts)
        if (nblk * (o[0] + o[1]) <= cus && 4 * o[0] <= ntiles) {
            *sd = o[0];
            *sw = o[1];
            return;
        }
    *sd = *sw = 1;
}

// launchers: one instantiation per slicing factor (dynamic LDS limit raised on the instance that is launched)
StageCfg plain_tile(const float* t) { return StageCfg{t, nullptr, nullptr, nullptr, nullptr, 0.0f, false, false, false}; }
// The dynamic-LDS limit of a kernel is raised once per process and size (per device: the attribute lives with the loaded code
// object; a second device in the same process raises its own on first use) instead of in front of every launch (measured: ~0.3 us
// each on this stack, twelve per step -- nothing a step's timing shows).
template <auto Kernel>      // (one table per kernel instantiation)
void raise_lds_limit(size_t lds) {
    static thread_local size_t granted[16] = {};
    howl_raise_lds(reinterpret_cast<const void*>(Kernel), lds, granted, "res8 kernel");   // refused: reported by HOWL_CHECK_LAUNCH
}

template <int MODE, int SLICES, int HALO = 0>
void launch_conv3x3_inst(int nblk, size_t lds, hipStream_t stream, const StageCfg& in, const float* in_stats, const float* wp,
                         const float* res, float* out, const float* xs, const float* xs_stats, float* part, int B, int H,
                         const BnFold& fold, const BwdFold& bfold, float* pool, const WFold& wf,
                         const StripGeom& sg = StripGeom{1, 1, 0}) {
    raise_lds_limit<conv3x3_mfma_kernel<MODE, SLICES, HALO>>(lds);
    hipLaunchKernelGGL((conv3x3_mfma_kernel<MODE, SLICES, HALO>), dim3(launch_blocks(nblk, SLICES)), dim3(CONV_THREADS), lds, stream, with_probe(in),
                       in_stats, wp, res, out, xs, xs_stats, part, pool, B, H, nblk, fold, bfold, wf, sg);
}
template <int MODE>
void launch_conv3x3(int slices, int nblk, size_t lds, hipStream_t stream, const StageCfg& in, const float* in_stats, const float* wp,
                    const float* res, float* out, const float* xs, const float* xs_stats, float* part, int B, int H,
                    const BnFold& fold, const BwdFold& bfold = BwdFold{}, float* pool = nullptr,
                    const WFold& wf = WFold{nullptr, 0, nullptr}, int halo = 0, const StripGeom& sg = StripGeom{1, 1, 0}) {
    if (halo == 2)      // row (and column) strips of a long map: StripGeom
        launch_conv3x3_inst<MODE, 1, 2>(nblk, lds, stream, in, in_stats, wp, res, out, xs, xs_stats, part, B, H, fold, bfold, pool, wf, sg);
    else if (halo == 1)      // strips of a wide map (NUM_MELS = 80): one workgroup per strip, no position slicing
        launch_conv3x3_inst<MODE, 1, 1>(nblk, lds, stream, in, in_stats, wp, res, out, xs, xs_stats, part, B, H, fold, bfold, pool, wf);
    else if (slices == 4)
        launch_conv3x3_inst<MODE, 4>(nblk, lds, stream, in, in_stats, wp, res, out, xs, xs_stats, part, B, H, fold, bfold, pool, wf);
    else if (slices == 2)
        launch_conv3x3_inst<MODE, 2>(nblk, lds, stream, in, in_stats, wp, res, out, xs, xs_stats, part, B, H, fold, bfold, pool, wf);
    else
        launch_conv3x3_inst<MODE, 1>(nblk, lds, stream, in, in_stats, wp, res, out, xs, xs_stats, part, B, H, fold, bfold, pool, wf);
}
template <int SW, int HALO = 0>
void launch_wgrad_inst(int nblk, size_t lds, hipStream_t stream, const StageCfg& zc, const BwdFold& bfold, const float* s_prev,
                       const float* in_stats, float* wpart, int B, int H, const StripGeom& sg = StripGeom{1, 1, 0}) {
    raise_lds_limit<wgrad_mfma_kernel<SW, HALO>>(lds);
    hipLaunchKernelGGL((wgrad_mfma_kernel<SW, HALO>), dim3(launch_blocks(nblk, SW)), dim3(CONV_THREADS), lds, stream,
                       WStage{with_probe(zc), s_prev, false}, in_stats, bfold, wpart, B, H, nblk, sg);
}
template <int SD, int SW, int HALO = 0>
void launch_pair_inst(int nblk, size_t lds, hipStream_t stream, const StageCfg& zc, const BwdFold& bfold, const float* wp, float* dx,
                      const float* xs, const float* xs_stats, float* spart, const float* s_prev, const float* in_stats, float* wpart,
                      int B, int H, const WFold& wf, const StripGeom& sg = StripGeom{1, 1, 0}) {
    raise_lds_limit<bwd_pair_kernel<SD, SW, HALO>>(lds);
    hipLaunchKernelGGL((bwd_pair_kernel<SD, SW, HALO>), dim3(launch_blocks(nblk, SD + SW)), dim3(CONV_THREADS), lds, stream, with_probe(zc), bfold, wp,
                       dx, xs, xs_stats, spart, s_prev, in_stats, wpart, B, H, nblk, wf, sg);
}
void launch_pair(int sd, int sw, int nblk, size_t lds, hipStream_t stream, const StageCfg& zc, const BwdFold& bfold, const float* wp,
                 float* dx, const float* xs, const float* xs_stats, float* spart, const float* s_prev, const float* in_stats,
                 float* wpart, int B, int H, const WFold& wf, int halo = 0, const StripGeom& sg = StripGeom{1, 1, 0}) {
    if (halo == 2)
        launch_pair_inst<1, 1, 2>(nblk, lds, stream, zc, bfold, wp, dx, xs, xs_stats, spart, s_prev, in_stats, wpart, B, H, wf, sg);
    else if (halo == 1)
        launch_pair_inst<1, 1, 1>(nblk, lds, stream, zc, bfold, wp, dx, xs, xs_stats, spart, s_prev, in_stats, wpart, B, H, wf);
    else if (sd == 4 && sw == 2)
        launch_pair_inst<4, 2>(nblk, lds, stream, zc, bfold, wp, dx, xs, xs_stats, spart, s_prev, in_stats, wpart, B, H, wf);
    else if (sd == 2 && sw == 2)
        launch_pair_inst<2, 2>(nblk, lds, stream, zc, bfold, wp, dx, xs, xs_stats, spart, s_prev, in_stats, wpart, B, H, wf);
    else if (sd == 2 && sw == 1)
        launch_pair_inst<2, 1>(nblk, lds, stream, zc, bfold, wp, dx, xs, xs_stats, spart, s_prev, in_stats, wpart, B, H, wf);
    else
        launch_pair_inst<1, 1>(nblk, lds, stream, zc, bfold, wp, dx, xs, xs_stats, spart, s_prev, in_stats, wpart, B, H, wf);
}

}  // namespace

extern "C" {

size_t howl_res8_workspace_bytes(int B, int T) { return howl_res8_workspace_bytes_mels(B, T, 40); }

size_t howl_res8_workspace_bytes_mels(int B, int T, int M) {
    const Strips sp = strips_for(B, T, M);
    return ws_layout(nullptr, nullptr, sp.Bv, sp.Hs, even_grid(conv_grid(sp.Bv), sp.ns));
}

size_t howl_res8_eval_workspace_bytes_mels(int B, int T, int M) {
    const Strips sp = strips_for(B, T, M);
    size_t eval_bytes = 0;
    ws_layout(nullptr, nullptr, sp.Bv, sp.Hs, even_grid(conv_grid(sp.Bv), sp.ns), &eval_bytes);
    return eval_bytes;
}

int howl_res8_row_strips(int T) { return strips_for(1, T, 40).nr; }

size_t howl_res8_saved_floats(int B, int T, int M) {
    const Strips sp = strips_for(B, T, M);
    return (size_t)sp.Bv * NMAP * sp.Hs * PW;
}

}  // extern "C"

namespace {
// howl_res8_fwd, optionally with the cross-entropy of the step in its last launch (labels != nullptr: see head_fwd_kernel)
int res8_fwd_impl(const HowlRes8Params* prm, const float* feat, long sb, long st, long sm, int B, int T, int M, int C,
                  int training, const HowlRes8Saved* sv, float* logits, void* ws, size_t ws_bytes, const long long* labels,
                  float* nll, float* dlogits, hipStream_t stream) {
    HOWL_REQUIRE(prm && feat && sv && logits && ws, "howl_res8_fwd: null pointer");
    HOWL_REQUIRE(labels == nullptr || (nll != nullptr && dlogits != nullptr && C <= HEAD_XC),
                 "howl_res8_fwd_xent: nll / dlogits missing or more than %d classes (C=%d)", HEAD_XC, C);
    HOWL_REQUIRE(mel_strips(M) > 0, "howl_res8_fwd: res8 pools (3,4) over 40 or 80 mel bins; got M=%d", M);
    const int Ht = T / 3;         // pooled rows of the whole map
    HOWL_REQUIRE(B >= 1 && Ht >= 1, "howl_res8_fwd: B=%d T=%d unsupported (T >= 3)", B, T);
    HOWL_REQUIRE(C >= 1, "howl_res8_fwd: C must be positive");
    // wide maps: every utterance is NS strips of 10 pooled columns (HaloSlot); long maps: NR row strips of H rows each (StripGeom);
    // every strip is a block of the (Bv, 45, H, 10) activations
    const Strips sp = strips_for(B, T, M);
    HOWL_REQUIRE(sp.hv_last >= 1 && sp.nr <= MAX_ROW_STRIPS, "howl_res8_fwd: T=%d frames unsupported (%d row strips)", T, sp.nr);
    const int NS = sp.ns, H = sp.Hs, Bv = sp.Bv, halo = sp.halo;
    const bool grid = halo == 2;
    const int G = even_grid(conv_grid(Bv), NS);
    Ws w;
    size_t need_eval = 0;
    const size_t need_train = ws_layout(&w, static_cast<char*>(ws), Bv, H, G, &need_eval);
    // eval mode touches nothing behind the pooled sums (howl_res8_eval_workspace_bytes_mels), and reads s[i-1], s[i-2] while it
    // writes s[i]: the caller may pass three activation buffers in rotation (s[i] = buffer i mod 3) instead of seven
    const size_t need = training ? need_train : need_eval;
    if (ws_bytes < need) {
        howl_set_error("howl_res8_fwd: workspace %zu < %zu bytes", ws_bytes, need);
        return HOWL_E_WORKSPACE;
    }
    const int P = H * PW;
    const int Pt = Ht * PW * NS;      // positions of one utterance's whole map
    HowlPtrs6 cw, rm, rv;
    for (int i = 0; i < 6; ++i) {
        cw.p[i] = const_cast<float*>(prm->conv_w[i]);
        rm.p[i] = prm->bn_running_mean[i];
        rv.p[i] = prm->bn_running_var[i];
    }
    if (!training) hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(6), dim3(64), 0, stream, rm, rv, sv->bn_stats);

    // conv0 sees an utterance (or, for row strips, its NR windows of 3 H frames with the clip's real frames on both sides)
    const int Bw = B * sp.nr, Tw = grid ? 3 * H : T;
    const size_t l0 = conv0_tile_floats(Tw, M) * sizeof(float);
    // 103 VGPRs and 15 KB of LDS: two workgroups per CU overlap one's tile load / stores with the other's MFMAs
    const int S0 = conv0_slices(Bw);
    const int G0 = Bw * S0 < 2 * howl_num_cus() ? Bw * S0 : 2 * howl_num_cus();
    {
        HowlProfScope prof("conv0_fwd", stream);
        const int npack = (2 * 6 * PACK_ELEMS + C0M_THREADS - 1) / C0M_THREADS;
#define HOWL_CONV0_FWD(NS_, EX_)                                                                                                  \
    hipLaunchKernelGGL((conv0_fwd_mfma_kernel<NS_, EX_>), dim3(G0 + npack), dim3(C0M_THREADS), l0, stream, feat, sb, st, sm,       \
                       prm->conv0_w, sv->s[0], sv->mask0, Bw, Tw, M, H, G0, cw, w.wp_fwd, w.wp_bwd, grid ? sp.nr : 1,             \
                       grid ? 3 * H : 0, grid ? T : 0, S0)
        if (grid && NS == 2)
            HOWL_CONV0_FWD(2, true);
        else if (grid)
            HOWL_CONV0_FWD(1, true);
        else if (NS == 2)
            HOWL_CONV0_FWD(2, false);
        else
            HOWL_CONV0_FWD(1, false);
#undef HOWL_CONV0_FWD
    }
    const size_t lc = conv_lds_bytes(H, grid);
    const double count = (double)B * (double)Pt;
    const int SL = halo != 0 ? 1 : conv_slices(G, H, howl_num_cus());
    // Training: layer i leaves its statistics as per-workgroup partials; layer i+1 folds them in its own prologue (BnFold),
    // so only the last layer needs the stand-alone finalize.  The partial buffers alternate between layers.
    for (int i = 1; i <= 6; ++i) {
        const bool even = (i % 2) == 0;
        const float* res = even ? sv->s[i - 2] : nullptr;
        float* part_out = training ? ((i & 1) ? w.part : w.part2) : (float*)nullptr;
        BnFold fold{};
        const float* in_stats = nullptr;
        if (i > 1) {
            if (training)
                fold = BnFold{(i & 1) ? w.part2 : w.part, G * SL, count, sv->bn_stats + (size_t)(i - 2) * 2 * CP,
                              HowlBnBuffers{prm->bn_running_mean[i - 2], prm->bn_running_var[i - 2], prm->bn_num_batches[i - 2]}};
            else
                in_stats = sv->bn_stats + (size_t)(i - 2) * 2 * CP;
        }
        {
            HowlProfScope prof("conv3x3_fwd", stream);
            launch_conv3x3<0>(SL, G, lc, stream, plain_tile(sv->s[i - 1]), in_stats, w.wp_fwd + (size_t)(i - 1) * 3 * KSTEPS * 64, res, sv->s[i],
                              nullptr, nullptr, part_out, Bv, H, fold, BwdFold{}, i == 6 ? w.pool : (float*)nullptr,
                              WFold{nullptr, 0, nullptr}, halo, sp.sg);
        }
    }
    // (training: the head folds BatchNorm 6's statistics itself -- no one-block finalize launch in between)
    const BnFold hfold = training ? BnFold{w.part2, G * SL, count, sv->bn_stats + (size_t)5 * 2 * CP,
                                           HowlBnBuffers{prm->bn_running_mean[5], prm->bn_running_var[5], prm->bn_num_batches[5]}}
                                  : BnFold{};
    // the spatial mean runs over all strips of an utterance: nr * ns blocks of 4 SL position groups, Pt positions
    hipLaunchKernelGGL(head_fwd_kernel, dim3(B < 1024 ? B : 1024), dim3(HEAD_THREADS), 0, stream, sv->s[6],
                       sv->bn_stats + (size_t)5 * 2 * CP, prm->out_w, prm->out_b, sv->pooled, logits, B, Pt, C, labels, nll, dlogits,
                       w.dpool, 1.0f / (float)B, (const float*)w.pool, 4 * SL, 4 * SL < (P + 15) / 16 ? 4 * SL : (P + 15) / 16, hfold,
                       sp.nr * NS);
    HOWL_CHECK_LAUNCH("howl_res8_fwd");
    return HOWL_OK;
}
}  // namespace

extern "C" {

int howl_res8_fwd(const HowlRes8Params* prm, const float* feat, long sb, long st, long sm, int B, int T, int M, int C,
                  int training, const HowlRes8Saved* sv, float* logits, void* ws, size_t ws_bytes, hipStream_t stream) {
    return res8_fwd_impl(prm, feat, sb, st, sm, B, T, M, C, training, sv, logits, ws, ws_bytes, nullptr, nullptr, nullptr, stream);
}

int howl_res8_fwd_xent(const HowlRes8Params* prm, const float* feat, long sb, long st, long sm, int B, int T, int M, int C,
                       const HowlRes8Saved* sv, const long long* labels, float* logits, float* nll, float* dlogits, void* ws,
                       size_t ws_bytes, hipStream_t stream) {
    HOWL_REQUIRE(labels && nll && dlogits, "howl_res8_fwd_xent: null pointer");
    return res8_fwd_impl(prm, feat, sb, st, sm, B, T, M, C, 1, sv, logits, ws, ws_bytes, labels, nll, dlogits, stream);
}

// ---- long inputs (eval mode) --------------------------------------------------------------------------------------------------
// The kernels keep one utterance's whole (45, H, 10) map in LDS, H <= 27 pooled rows (83 frames).  A longer clip is cut into
// windows of 27 pooled rows that overlap by 14: the six 3x3 convolutions (and conv0's own frame) spread a window's zero
// padding 7 rows inwards, so a window's rows [7, 20) -- [0, 20) for the first, [7, 27) for the last -- are exactly what the
// unbounded convolution stack computes, and those ranges tile the clip.  Inference statistics are per channel constants,
// so nothing couples the windows except the final spatial mean.  (Training on such windows would need BatchNorm batch
// statistics over the de-duplicated rows: not offered; the reference's training windows are <= 1 s.)
namespace {
constexpr int WIN_H = MAX_H, WIN_MARGIN = 7, WIN_STEP = WIN_H - 2 * WIN_MARGIN;   // 27, 7, 13
int long_windows(int H) { return H <= WIN_H ? 1 : (H - WIN_H + WIN_STEP - 1) / WIN_STEP + 1; }
}  // namespace

size_t howl_res8_long_workspace_bytes(int B, int T) { return howl_res8_long_workspace_bytes_mels(B, T, 40); }

size_t howl_res8_long_workspace_bytes_mels(int B, int T, int M) {
    const int H = T / 3, nw = long_windows(H);
    const size_t Bv = (size_t)B * nw * (mel_strips(M) > 0 ? mel_strips(M) : 1);
    // three rotating activation maps + eval statistics + the regular workspace of the virtual batch
    static_assert((6 * 2 * CP * sizeof(float)) % 256 == 0, "statistics block keeps the 256-byte alignment");
    return 3 * (((Bv * NMAP * WIN_H * PW * sizeof(float)) + 255) / 256 * 256) + 6 * 2 * CP * sizeof(float) + 256 +
           ws_layout(nullptr, nullptr, (int)Bv, WIN_H, even_grid(conv_grid((int)Bv), mel_strips(M)));
}

int howl_res8_fwd_long(const HowlRes8Params* prm, const float* feat, long sb, long st, long sm, int B, int T, int M, int C,
                       float* logits, void* ws, size_t ws_bytes, hipStream_t stream) {
    HOWL_REQUIRE(prm && feat && logits && ws, "howl_res8_fwd_long: null pointer");
    const int NS = mel_strips(M);
    HOWL_REQUIRE(NS > 0, "howl_res8_fwd_long: res8 pools (3,4) over 40 or 80 mel bins; got M=%d", M);
    const int H = T / 3;
    HOWL_REQUIRE(B >= 1 && H > WIN_H && C >= 1, "howl_res8_fwd_long: for T > 83 frames (got B=%d T=%d); shorter inputs use howl_res8_fwd", B, T);
    const int nw = long_windows(H);
    HOWL_REQUIRE(nw <= MAX_WINDOWS, "howl_res8_fwd_long: T=%d needs %d windows (max %d)", T, nw, MAX_WINDOWS);
    if (ws_bytes < howl_res8_long_workspace_bytes_mels(B, T, M)) {
        howl_set_error("howl_res8_fwd_long: workspace %zu < %zu bytes", ws_bytes, howl_res8_long_workspace_bytes_mels(B, T, M));
        return HOWL_E_WORKSPACE;
    }
    // T % 3 trailing frames take no part in the pooling but are conv0's neighbours of the clip's last frame: every window
    // reads them too (inner windows simply see real samples there instead of padding, inside their discarded margin)
    // Bw windows, each NS strips of 10 pooled columns (wide maps: HaloSlot): Bv blocks of (45, WIN_H, 10)
    const int Bw = B * nw, Bv = Bw * NS, Tw = 3 * WIN_H + T % 3, P = WIN_H * PW;
    const bool halo = NS > 1;
    const size_t act = ((size_t)Bv * NMAP * P * sizeof(float) + 255) / 256 * 256;
    char* base = static_cast<char*>(ws);
    float* buf[3] = {reinterpret_cast<float*>(base), reinterpret_cast<float*>(base + act), reinterpret_cast<float*>(base + 2 * act)};
    float* stats = reinterpret_cast<float*>(base + 3 * act);
    Ws w;
    const int G = even_grid(conv_grid(Bv), NS);
    ws_layout(&w, base + 3 * act + 6 * 2 * CP * sizeof(float) + 256, Bv, WIN_H, G);
    HowlWinRows rows;
    for (int i = 0; i < nw; ++i) {
        const int a = i * WIN_STEP < H - WIN_H ? i * WIN_STEP : H - WIN_H;          // first pooled row of window i
        const int a_next = (i + 1) * WIN_STEP < H - WIN_H ? (i + 1) * WIN_STEP : H - WIN_H;
        const int g_lo = i == 0 ? 0 : a + WIN_MARGIN;                              // clip rows [g_lo, g_hi) come from window i
        const int g_hi = i == nw - 1 ? H : a_next + WIN_MARGIN;
        rows.lo[i] = g_lo - a;
        rows.hi[i] = g_hi - a;
    }
    HowlPtrs6 cw, rm, rv;
    for (int i = 0; i < 6; ++i) {
        cw.p[i] = const_cast<float*>(prm->conv_w[i]);
        rm.p[i] = prm->bn_running_mean[i];
        rv.p[i] = prm->bn_running_var[i];
    }
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(6), dim3(64), 0, stream, rm, rv, stats);
    const size_t l0 = conv0_tile_floats(Tw, M) * sizeof(float);
    const int G0 = Bw < 2 * howl_num_cus() ? Bw : 2 * howl_num_cus();
    const int npack = (2 * 6 * PACK_ELEMS + C0M_THREADS - 1) / C0M_THREADS;
    if (halo)
        hipLaunchKernelGGL(conv0_fwd_mfma_kernel<2>, dim3(G0 + npack), dim3(C0M_THREADS), l0, stream, feat, sb, st, sm, prm->conv0_w,
                           buf[0], (unsigned short*)nullptr, Bw, Tw, M, WIN_H, G0, cw, w.wp_fwd, w.wp_bwd, nw, 3 * WIN_STEP,
                           3 * (H - WIN_H), 1);
    else
        hipLaunchKernelGGL(conv0_fwd_mfma_kernel<1>, dim3(G0 + npack), dim3(C0M_THREADS), l0, stream, feat, sb, st, sm, prm->conv0_w,
                           buf[0], (unsigned short*)nullptr, Bw, Tw, M, WIN_H, G0, cw, w.wp_fwd, w.wp_bwd, nw, 3 * WIN_STEP,
                           3 * (H - WIN_H), 1);
    const size_t lc = conv_lds_bytes(WIN_H);
    const int SL = halo ? 1 : conv_slices(G, WIN_H, howl_num_cus());
    // x_i lives in buf[cur]; even layers add the map two layers back (kept in buf[skip])
    int cur = 0, skip = 0;
    for (int i = 1; i <= 6; ++i) {
        const bool even = (i % 2) == 0;
        int out = 0;
        while (out == cur || out == skip) ++out;
        launch_conv3x3<0>(SL, G, lc, stream, plain_tile(buf[cur]), i == 1 ? (const float*)nullptr : (const float*)(stats + (size_t)(i - 2) * 2 * CP),
                          w.wp_fwd + (size_t)(i - 1) * 3 * KSTEPS * 64, even ? (const float*)buf[skip] : (const float*)nullptr, buf[out],
                          nullptr, nullptr, nullptr, Bv, WIN_H, BnFold{}, BwdFold{}, nullptr, WFold{nullptr, 0, nullptr}, halo);
        if (even) skip = out;      // s_i (i even) is the next residual source; s_0 is the first one
        cur = out;
    }
    hipLaunchKernelGGL(head_fwd_windows_kernel, dim3(B < 1024 ? B : 1024), dim3(256), 0, stream, (const float*)buf[cur],
                       (const float*)(stats + (size_t)5 * 2 * CP), prm->out_w, prm->out_b, logits, B, nw, WIN_H, rows,
                       1.0f / ((float)H * PW * NS), C, NS);
    HOWL_CHECK_LAUNCH("howl_res8_fwd_long");
    return HOWL_OK;
}

// part 0: the whole backward pass.  Data-parallel steps call it in two parts so that the gradient all-reduce of everything
// but conv0 runs under conv0's weight gradient: part 1 = head + layers 6..1 (data and weight gradients) + the fold of the six
// layers' weight-gradient partials -- after it gr->conv_w[0..5], gr->out_w, gr->out_b are final; part 2 = conv0's weight
// gradient and its fold (gr->conv0_w).  The two parts of a pass must be called in order with identical arguments.
}  // extern "C"

namespace {
// howl_res8_bwd_part; with nll != nullptr the pass follows howl_res8_fwd_xent: the pooled gradient is already in the workspace
// and the batch mean of nll goes to `loss` (one more block of the head's parameter-gradient launch)
int res8_bwd_impl(const HowlRes8Params* prm, const float* feat, long sb, long st, long sm, int B, int T, int M, int C,
                  const HowlRes8Saved* sv, const float* dlogits, const HowlRes8Grads* gr, void* ws, size_t ws_bytes,
                  int part, const float* nll, float* loss, hipStream_t stream, const HowlAdamW* adamw = nullptr) {
    HOWL_REQUIRE(prm && feat && sv && dlogits && gr && ws, "howl_res8_bwd: null pointer");
    HOWL_REQUIRE(adamw == nullptr || (part == 0 && adamw->p && adamw->g && adamw->m && adamw->v && adamw->n >= 1 && adamw->step >= 1),
                 "howl_res8_bwd: HowlAdamW needs the whole pass (part 0) and complete buffers");
    HOWL_REQUIRE(part >= 0 && part <= 2, "howl_res8_bwd_part: part must be 0 (all), 1 or 2");
    const bool run_layers = part != 2, run_conv0 = part != 1;
    HOWL_REQUIRE(mel_strips(M) > 0, "howl_res8_bwd: M must be 40 or 80 (got %d)", M);
    const int Ht = T / 3;
    HOWL_REQUIRE(B >= 1 && Ht >= 1, "howl_res8_bwd: B=%d T=%d unsupported", B, T);
    const Strips sp = strips_for(B, T, M);      // column strips (HaloSlot) x row strips (StripGeom), each a block of the activations
    HOWL_REQUIRE(sp.hv_last >= 1 && sp.nr <= MAX_ROW_STRIPS, "howl_res8_bwd: T=%d frames unsupported (%d row strips)", T, sp.nr);
    const int NS = sp.ns, H = sp.Hs, Bv = sp.Bv, halo = sp.halo;
    const bool grid = halo == 2;
    const int Pt = Ht * PW * NS;      // positions of one utterance's whole map
    const int G = even_grid(conv_grid(Bv), NS);
    Ws w;
    const size_t need = ws_layout(&w, static_cast<char*>(ws), Bv, H, G);
    if (ws_bytes < need) {
        howl_set_error("howl_res8_bwd: workspace %zu < %zu bytes", ws_bytes, need);
        return HOWL_E_WORKSPACE;
    }
    const int P = H * PW;
    const double count = (double)B * (double)Pt;
    const size_t act = (size_t)Bv * NMAP * P;
    HOWL_REQUIRE(act / 2 < (size_t)1 << 31, "howl_res8_bwd: B=%d too large for the 32-bit element index of the elementwise pass", B);
    int eg = (int)((act / 4 + BRB_THREADS - 1) / BRB_THREADS);      // one 16-byte quad per thread and trip
    if (eg < 1) eg = 1;
    if (eg > 2 * howl_num_cus()) eg = 2 * howl_num_cus();

    if (run_layers) {
        if (nll == nullptr)
            hipLaunchKernelGGL(head_bwd_pool_kernel, dim3((B * CP + 255) / 256), dim3(256), 0, stream, dlogits, prm->out_w, w.dpool,
                               B, C);
        hipLaunchKernelGGL(head_bwd_param_kernel, dim3(C + (nll != nullptr ? 2 : 1)), dim3(1024), 0, stream, dlogits, sv->pooled,
                           w.dpool, gr->out_w, gr->out_b, w.m12, B, C, Pt, nll, loss);
    }
    const size_t lc = conv_lds_bytes(H, grid);
    const size_t lw = wgrad_lds_bytes(H);
    const size_t lp = lc > lw ? lc : lw;
    const char* pair_env = getenv("HOWL_RES8_BWD_PAIR");
    const bool merged = !(pair_env != nullptr && pair_env[0] == '0');
    // HOWL_RES8_BWD_FUSED=0: the elementwise BatchNorm / ReLU backward as its own launch per layer (bn_relu_bwd_kernel writes
    // dz_i, the pair stages it as it is) -- the reference point of the tests; default: built inside the pair's staging
    const char* fused_env = getenv("HOWL_RES8_BWD_FUSED");
    const bool fused = halo != 0 || !(fused_env != nullptr && fused_env[0] == '0');   // (strips: the fused staging only)
    // dgrad and wgrad side by side: half the CUs each
    const int half = howl_num_cus() / 2 > 0 ? howl_num_cus() / 2 : 1;
    const int Gh = even_grid(Bv < half ? Bv : half, NS);
    const size_t wpart_stride = (size_t)Gh * CP * WNCOL;
    int SD = 1, SW = 1;
    if (halo == 0) pair_slices(Gh, H, &SD, &SW);
    // the fold of a layer's weight-gradient partials rides in the NEXT pair launch when that launch has enough data-gradient
    // workgroups to spread the 9,984 column pairs thin (a single utterance's four workgroups would walk 26 trips of two barriers
    // each: +60 us at batch 1); small batches keep the one reduction launch at the end
    const bool fold_in_pair = Gh * SD >= 64;
    float* dx_cur = nullptr;      // gradient w.r.t. the BN output of layer i (nullptr: broadcast of dpool)
    float* dx_next = w.bufa;
    float* ds_prev = nullptr;     // ds_{i+2}
    float* ds_free = w.dsa;
    for (int i = 6; i >= 1; --i) {
        const bool even = (i % 2) == 0;
        const float* stats_i = sv->bn_stats + (size_t)(i - 1) * 2 * CP;
        float* ds_out = even ? ds_free : nullptr;
        float* dz = even ? w.dz : w.dz2;
        // the BatchNorm-backward statistics travel as partials from the data gradient of layer i+1 to the consumer of dx_i; the
        // fused pair reads them in its prologue while its own data-gradient workgroups write theirs at their end: two buffers
        float* part_in = fused ? ((i & 1) ? w.part2 : w.part) : w.part;
        float* part_out = fused ? ((i & 1) ? w.part : w.part2) : w.part;
        // layer 6 takes its two means from the head (m12); the others fold the partials of the data gradient above them
        const BwdFold bfold{stats_i, w.m12, i == 6 ? (const float*)nullptr : (const float*)part_in, Gh * SD, count};
        StageCfg zc = plain_tile(dz);
        if (fused)
            zc = StageCfg{dx_cur, sv->s[i], even ? (const float*)ds_prev : (const float*)nullptr, ds_out, w.dpool,
                          1.0f / (float)Pt, true, even, false};
        else if (run_layers)
            hipLaunchKernelGGL(bn_relu_bwd_kernel, dim3(eg), dim3(BRB_THREADS), 0, stream, (const float*)dx_cur, w.dpool, sv->s[i],
                               stats_i, w.m12, bfold.part, Gh * SD, count, even ? (const float*)ds_prev : (const float*)nullptr,
                               even ? 1 : 0, ds_out, dz, B, P);
        if (even) {
            float* t = ds_prev ? ds_prev : w.dsb;
            ds_prev = ds_out;
            ds_free = t;
        }
        // data gradient: dx_{i-1} (w.r.t. the normalised input of layer i), with BN_{i-1} backward statistics;
        // weight gradient of layer i: input x_{i-1} = BN_{i-1}(s_{i-1}) (identity for i = 1)
        const float* in_stats = (i == 1) ? nullptr : sv->bn_stats + (size_t)(i - 2) * 2 * CP;
        const bool need_stats = i > 1;
        const float* wpb = w.wp_bwd + (size_t)(i - 1) * 3 * KSTEPS * 64;
        // layer 1 needs no statistics (conv0 has no BatchNorm in front): its data gradient takes the skip gradient ds_2 as an
        // addend instead, so that dx_0 + ds_2 = the gradient of conv0's pooled output leaves the launch as ONE map
        const float* xs = need_stats ? sv->s[i - 1] : (const float*)ds_prev;
        const float* xs_st = need_stats ? in_stats : (const float*)nullptr;
        float* spart = need_stats ? part_out : (float*)nullptr;
        float* wpart = w.wpart + (size_t)(i - 1) * wpart_stride;
        // the partials of layer i+1's weight gradient (previous launch) are folded by this launch's data-gradient workgroups
        const WFold wf = (fold_in_pair && i < 6) ? WFold{w.wpart + (size_t)i * wpart_stride, Gh, gr->conv_w[i]} : WFold{nullptr, 0, nullptr};
        if (!run_layers) {
            // part 2 only replays the buffer rotation of the loop
        } else if (merged) {
            HowlProfScope prof("bwd_pair", stream);
            launch_pair(SD, SW, Gh, lp, stream, zc, bfold, wpb, dx_next, xs, xs_st, spart, sv->s[i - 1], in_stats, wpart, Bv, H, wf, halo, sp.sg);
        } else {
            {
                HowlProfScope prof("conv3x3_dgrad", stream);
                launch_conv3x3<1>(SD, Gh, lc, stream, zc, nullptr, wpb, nullptr, dx_next, xs, xs_st, spart, Bv, H, BnFold{}, bfold, nullptr, wf, halo, sp.sg);
            }
            HowlProfScope prof("wgrad", stream);
            StageCfg zw = zc;
            zw.ds = nullptr;
            if (halo == 2)
                launch_wgrad_inst<1, 2>(Gh, lw, stream, zw, bfold, sv->s[i - 1], in_stats, wpart, Bv, H, sp.sg);
            else if (halo == 1)
                launch_wgrad_inst<1, 1>(Gh, lw, stream, zw, bfold, sv->s[i - 1], in_stats, wpart, Bv, H);
            else if (SW == 2)
                launch_wgrad_inst<2>(Gh, lw, stream, zw, bfold, sv->s[i - 1], in_stats, wpart, B, H);
            else
                launch_wgrad_inst<1>(Gh, lw, stream, zw, bfold, sv->s[i - 1], in_stats, wpart, B, H);
        }
        dx_cur = dx_next;
        dx_next = (dx_next == w.bufa) ? w.bufb : w.bufa;
    }
    // conv0: dy0 = dx_0 + ds_2 (skip into s_2 = y_2 + y_0), summed by layer 1's data gradient (ConvEpilogue::xadd)
    HOWL_REQUIRE(sv->mask0 != nullptr, "howl_res8_bwd: saved->mask0 is required");
    HowlPtrs6 gw;
    for (int i = 0; i < 6; ++i) gw.p[i] = gr->conv_w[i];
    const int S0 = conv0_slices(Bv);
    const int G0w = Bv * S0 < howl_num_cus() ? Bv * S0 : howl_num_cus();   // conv0's weight-gradient grid: one partial row each
    // layers 2..6 were folded inside the pair launches (WFold); layer 1's partials and conv0's remain
    const RowsAdamW no_opt{nullptr, nullptr, nullptr, nullptr, HowlAdamWCoef{}, 0, 0, 0};
    if (part == 1)      // the six layers' weight gradients are final before conv0's is even started
        hipLaunchKernelGGL(reduce_rows_all_kernel, dim3((CP * WNCOL + 63) / 64, fold_in_pair ? 1 : 6), dim3(1024), 0, stream,
                           (const float*)w.wpart, wpart_stride, Gh, gw, (const float*)w.c0part, G0w, gr->conv0_w, 0, 0,
                           fold_in_pair ? 1 : 6, no_opt);
    if (run_conv0) {
        {
        HowlProfScope prof("conv0_wgrad", stream);
        const int Tw = grid ? 3 * H : T;      // frames of one row strip (its window of the clip), or the utterance
        const size_t l0w = ((size_t)(Tw + 2) * (M + 4) + 16) * sizeof(float);
#define HOWL_CONV0_WGRAD(NS_, EX_)                                                                                                \
    hipLaunchKernelGGL((conv0_wgrad_valu_kernel<NS_, EX_>), dim3(G0w), dim3(C0G_THREADS), l0w, stream, feat, sb, st, sm,           \
                       (const unsigned short*)sv->mask0, (const float*)dx_cur, (const float*)nullptr, w.c0part, B, Tw, M, H, S0,  \
                       sp.sg, T)
        if (grid && NS == 2)
            HOWL_CONV0_WGRAD(2, true);
        else if (grid)
            HOWL_CONV0_WGRAD(1, true);
        else if (NS == 2)
            HOWL_CONV0_WGRAD(2, false);
        else
            HOWL_CONV0_WGRAD(1, false);
#undef HOWL_CONV0_WGRAD
        }
        if (part == 0) {    // rows {layer 1, conv0} of the reduction (small batches: all six layers and conv0)
            // ... and, on a single replica, the optimiser step (HowlAdamW): folded rows are updated as they are written, one more
            // row of blocks takes the parameters whose gradients were final before this launch.  Only for gradient pointers
            // that ARE the flat buffer in hot_parameters() order (conv0, conv1..6, output.weight, output.bias); otherwise the
            // step runs as its own launch behind this one.
            const int nrows = fold_in_pair ? 2 : 7;
            RowsAdamW opt = no_opt;
            bool own_launch = adamw != nullptr;
            if (adamw != nullptr && getenv("HOWL_NO_FOLD_ADAMW") == nullptr) {
                const float* g0 = adamw->g;
                bool flat = gr->conv0_w == g0 && gr->out_b == gr->out_w + (size_t)NMAP * C &&
                            adamw->n == (size_t)NMAP * 9 + (size_t)6 * NMAP * NMAP * 9 + (size_t)NMAP * C + (size_t)C &&
                            gr->out_w == g0 + (size_t)NMAP * 9 + (size_t)6 * NMAP * NMAP * 9;
                for (int i = 0; i < 6; ++i) flat = flat && gr->conv_w[i] == g0 + (size_t)NMAP * 9 + (size_t)i * NMAP * NMAP * 9;
                if (flat) {
                    const double bc1 = 1.0 - pow((double)adamw->beta1, (double)adamw->step);
                    const double bc2 = 1.0 - pow((double)adamw->beta2, (double)adamw->step);
                    opt = RowsAdamW{adamw->p, g0, adamw->m, adamw->v,
                                    HowlAdamWCoef{adamw->lr, adamw->beta1, adamw->beta2, adamw->eps, adamw->weight_decay, (float)bc1,
                                                  (float)sqrt(bc2), adamw->grad_scale},
                                    (long)((fold_in_pair ? gr->conv_w[1] : gr->out_w) - g0), (long)adamw->n, 1};
                    own_launch = false;
                }
            }
            hipLaunchKernelGGL(reduce_rows_all_kernel, dim3((CP * WNCOL + 63) / 64, nrows + (opt.on ? 1 : 0)), dim3(1024), 0, stream,
                               (const float*)w.wpart, wpart_stride, Gh, gw, (const float*)w.c0part, G0w, gr->conv0_w, 0,
                               fold_in_pair ? 5 : 0, nrows, opt);
            if (own_launch) {
                const int rc = howl_adamw_step(adamw->p, adamw->g, adamw->m, adamw->v, adamw->n, adamw->lr, adamw->beta1, adamw->beta2,
                                               adamw->eps, adamw->weight_decay, adamw->step, adamw->grad_scale, stream);
                if (rc != HOWL_OK) return rc;
            }
        } else {
            hipLaunchKernelGGL(reduce_rows_all_kernel, dim3((NMAP * 9 + 63) / 64, 1), dim3(1024), 0, stream,
                               (const float*)w.wpart, wpart_stride, Gh, gw, (const float*)w.c0part, G0w, gr->conv0_w, 6, 0, 1, no_opt);
        }
    }
    HOWL_CHECK_LAUNCH("howl_res8_bwd");
    return HOWL_OK;
}

}  // namespace

extern "C" {

int howl_res8_bwd_part(const HowlRes8Params* prm, const float* feat, long sb, long st, long sm, int B, int T, int M, int C,
                       const HowlRes8Saved* sv, const float* dlogits, const HowlRes8Grads* gr, void* ws, size_t ws_bytes,
                       int part, hipStream_t stream) {
    return res8_bwd_impl(prm, feat, sb, st, sm, B, T, M, C, sv, dlogits, gr, ws, ws_bytes, part, nullptr, nullptr, stream);
}

int howl_res8_bwd(const HowlRes8Params* prm, const float* feat, long sb, long st, long sm, int B, int T, int M, int C,
                  const HowlRes8Saved* sv, const float* dlogits, const HowlRes8Grads* gr, void* ws, size_t ws_bytes,
                  hipStream_t stream) {
    return res8_bwd_impl(prm, feat, sb, st, sm, B, T, M, C, sv, dlogits, gr, ws, ws_bytes, 0, nullptr, nullptr, stream);
}

int howl_res8_bwd_xent(const HowlRes8Params* prm, const float* feat, long sb, long st, long sm, int B, int T, int M, int C,
                       const HowlRes8Saved* sv, const float* dlogits, const float* nll, float* loss, const HowlRes8Grads* gr,
                       void* ws, size_t ws_bytes, int part, const HowlAdamW* adamw, hipStream_t stream) {
    HOWL_REQUIRE(nll && loss, "howl_res8_bwd_xent: null pointer");
    return res8_bwd_impl(prm, feat, sb, st, sm, B, T, M, C, sv, dlogits, gr, ws, ws_bytes, part, nll, loss, stream, adamw);
}

int howl_xent_fwd_bwd(const float* logits, const long long* labels, int B, int C, float* loss, float* dlogits,
                      hipStream_t stream) {
    HOWL_REQUIRE(logits && labels && loss, "howl_xent_fwd_bwd: null pointer");
    HOWL_REQUIRE(B >= 1 && C >= 1, "howl_xent_fwd_bwd: bad shape");
    hipLaunchKernelGGL(xent_kernel, dim3(1), dim3(1024), 0, stream, logits, labels, B, C, loss, dlogits);
    HOWL_CHECK_LAUNCH("howl_xent_fwd_bwd");
    return HOWL_OK;
}

int howl_adamw_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int step, float grad_scale, hipStream_t stream) {
    HOWL_REQUIRE(p && g && m && v, "howl_adamw_step: null pointer");
    HOWL_REQUIRE(step >= 1, "howl_adamw_step: step counts from 1");
    if (n == 0) return HOWL_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, g, m, v, n,
                       HowlAdamWCoef{lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale});
    HOWL_CHECK_LAUNCH("howl_adamw_step");
    return HOWL_OK;
}

}  // extern "C"
