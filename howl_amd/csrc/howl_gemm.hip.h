// Shared dense building blocks of the LSTM and MobileNet paths (included by lstm.hip and mobilenet.hip; every
// translation unit gets its own copy in an anonymous namespace): the generic fp32 MFMA GEMM with row maps and
// split-K, deterministic slab / column sums, and their host-side launch helpers.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "howl_common.hip.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// Generic fp32 MFMA GEMM:  C[m][n] = sum_k A(m,k) * B(k,n)  (+ bias[n]) (ReLU)
//   A(m,k) = a[am(m) + k*a_ks], B(k,n) = b[bk(k) + n*b_ns]; am/bk are two-level affine maps
//   off(x) = (x / inner) * s_outer + (x % inner) * s_inner, which lets (B,T,.) tensors with per-utterance gaps
//   (e.g. hseq (B,T+1,128)) be used as row sets without copies.  Split-K over gridDim.z writes partial slabs.
// 64x64 tile, BK = 16, 4 waves in a 2x2 grid, 2x2 16x16x4 MFMA tiles per wave.
// ---------------------------------------------------------------------------------------------------------
struct RowMap {
    int inner;
    long s_outer, s_inner;
};
__device__ __forceinline__ long rmap(const RowMap& r, int x) {
    if (r.inner >= (1 << 30)) return (long)x * r.s_inner;   // plain stride (uniform branch): no integer division
    return (long)(x / r.inner) * r.s_outer + (long)(x % r.inner) * r.s_inner;
}

constexpr int GT = 64, GK = 16, GLD = 80;  // tile edge, k depth, LDS row stride (80 = 16 mod 32: conflict-free frags)

template <bool A_MAJOR_IS_K>  // true: A's unit-stride index is k (row-major [m][k]); false: unit stride is m ([k][m])
__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ a, RowMap am, long a_ks, RowMap ak,
                                                   const float* __restrict__ b, RowMap bk, long b_ns, int M, int N, int K,
                                                   int k_per_split, const float* __restrict__ bias, int relu,
                                                   float* __restrict__ c, long c_ms, long c_split_stride) {
    __shared__ float As[GK * GLD];
    __shared__ float Bs[GK * GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
    const int kbeg = blockIdx.z * k_per_split;
    const int kend = min(K, kbeg + k_per_split);
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = {0.0f, 0.0f, 0.0f, 0.0f};

    for (int k0 = kbeg; k0 < kend; k0 += GK) {
        // stage A tile (64 m x 16 k) and B tile (16 k x 64 n); thread mapping follows the unit-stride index.  The eight
        // loads of a thread go out together from clamped addresses and are zeroed afterwards: guarded loads were compiled
        // into load / s_waitcnt vmcnt(0) pairs, eight dependent memory round trips per K tile.
        float va[4], vb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int mm = A_MAJOR_IS_K ? (tid >> 4) + 16 * j : tid & 63;
            const int kk = A_MAJOR_IS_K ? tid & 15 : (tid >> 6) + 4 * j;
            const int m = min(m0 + mm, M - 1), k = min(k0 + kk, kend - 1);
            va[j] = a[rmap(am, m) + (A_MAJOR_IS_K ? (long)k * a_ks : rmap(ak, k))];
            const int nn = tid & 63, kb = (tid >> 6) + 4 * j;
            const int n = min(n0 + nn, N - 1), k2 = min(k0 + kb, kend - 1);
            vb[j] = b[rmap(bk, k2) + (long)n * b_ns];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int mm = A_MAJOR_IS_K ? (tid >> 4) + 16 * j : tid & 63;
            const int kk = A_MAJOR_IS_K ? tid & 15 : (tid >> 6) + 4 * j;
            As[kk * GLD + mm] = (m0 + mm < M && k0 + kk < kend) ? va[j] : 0.0f;
            const int nn = tid & 63, kb = (tid >> 6) + 4 * j;
            Bs[kb * GLD + nn] = (n0 + nn < N && k0 + kb < kend) ? vb[j] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < GK / 4; ++ks) {
            const int kr = 4 * ks + (lane >> 4);
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = As[kr * GLD + 32 * wr + 16 * i + (lane & 15)];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = Bs[kr * GLD + 32 * wc + 16 * j + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    float* cz = c + (long)blockIdx.z * c_split_stride;
    // Bias and ReLU are applied to all 16 results in registers BEFORE the first store.  A bias load inside the guarded
    // store was 16 dependent round trips per lane; and even with the two values loaded up front, their first use inside
    // each guarded block put an s_waitcnt vmcnt there -- vmcnt also counts stores on gfx9, so every store waited for the
    // previous one to complete.
    if (bias != nullptr) {
        float bv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[j] = bias[min(n0 + 32 * wc + 16 * j + (lane & 15), N - 1)];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += bv[j];
    }
    if (relu) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = fmaxf(acc[i][j][r], 0.0f);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + 32 * wc + 16 * j + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 32 * wr + 16 * i + 4 * (lane >> 4) + r;
                if (m < M && n < N) cz[(long)m * c_ms + n] = acc[i][j][r];
            }
        }
}

// (Measured alternative, round 3: 128 x 128 / 128 x 64 block tiles with 16-byte LDS fragment reads and the same two-tile
// prefetch ran the six GEMMs of the seq-lstm step 3-20 % SLOWER than this 64 x 64 kernel -- 19456 rows give 1216 or 2432 small
// blocks, which balance over 256 CUs and keep 4x more loads in flight; with the MFMAs removed the kernel still takes 55-65 %
// of its time: it is bound by load latency / memory-level parallelism, not by LDS or matrix issue.  tools/gemm_variants.py)
// Fast path of the same GEMM for plain strided matrices whose unit-stride extents are multiples of 4 floats (every
// MobileNet 1x1 convolution and weight gradient, the LSTM projections): operands move as 16-byte vectors, one per
// thread and tile, and the next K tile is requested before the MFMAs of the current one (register double buffering).
//   A_UNIT_K: A(m,k) = a[amap(m) + k]   else a[amap(k) + m]      (amap / bmap: two-level row maps, strides multiples of 4)
//   B_UNIT_K: B(k,n) = b[bmap(n) + k]   else b[bmap(k) + n]
template <bool A_UNIT_K, bool B_UNIT_K, bool KMAP_LIN>
__global__ __launch_bounds__(256) void gemm_vec_kernel(const float* __restrict__ a, RowMap amap, const float* __restrict__ b,
                                                       RowMap bmap, int M, int N, int K, int k_per_split,
                                                       const float* __restrict__ bias, int relu, float* __restrict__ c,
                                                       long c_ms, long c_split_stride) {
    __shared__ float As[GK * GLD];
    __shared__ float Bs[GK * GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
    const int kbeg = blockIdx.z * k_per_split;
    const int kend = min(K, kbeg + k_per_split);
    // this thread's 4-float piece of each operand tile: unit-k -> (row = tid/4, k4 = tid%4); unit-row -> (k = tid/16, r4 = tid%16)
    const int a_r = A_UNIT_K ? tid >> 2 : (tid & 15) * 4, a_k = A_UNIT_K ? (tid & 3) * 4 : tid >> 4;
    const int b_r = B_UNIT_K ? tid >> 2 : (tid & 15) * 4, b_k = B_UNIT_K ? (tid & 3) * 4 : tid >> 4;
    // operand pieces are loaded unconditionally from clamped coordinates and zeroed when they are staged (ok_a / ok_b): a
    // guarded load would make the compiler wait for every outstanding load (vmcnt(0)) instead of only the oldest tile.
    // Row maps indexed by this thread's fixed row are evaluated once; maps indexed by k are a plain multiply when
    // KMAP_LIN (compile time: a run-time choice inside the loads had the same effect as a guard).
    auto kmap = [&](const RowMap& r, int k) -> long {
        if (KMAP_LIN) return (long)k * r.s_inner;
        return (long)(k / r.inner) * r.s_outer + (long)(k % r.inner) * r.s_inner;
    };
    const long a_fix = A_UNIT_K ? rmap(amap, min(m0 + a_r, M - 1)) : (long)min(m0 + a_r, M - 4);
    const long b_fix = B_UNIT_K ? rmap(bmap, min(n0 + b_r, N - 1)) : (long)min(n0 + b_r, N - 4);
    auto fetch_a = [&](int k0) -> float4 {
        const int k = min(k0 + a_k, A_UNIT_K ? kend - 4 : kend - 1);
        return *reinterpret_cast<const float4*>(A_UNIT_K ? a + a_fix + k : a + kmap(amap, k) + a_fix);
    };
    auto fetch_b = [&](int k0) -> float4 {
        const int k = min(k0 + b_k, B_UNIT_K ? kend - 4 : kend - 1);
        return *reinterpret_cast<const float4*>(B_UNIT_K ? b + b_fix + k : b + kmap(bmap, k) + b_fix);
    };
    auto ok_a = [&](int k0) { return m0 + a_r < M && k0 + a_k < kend; };
    auto ok_b = [&](int k0) { return n0 + b_r < N && k0 + b_k < kend; };
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = {0.0f, 0.0f, 0.0f, 0.0f};
    // Two K tiles in flight: tile k0 sits in one register pair while tiles k0+16 AND k0+32 are on their way (the loop is
    // unrolled by two so that the pairs swap roles without copies).  With one tile in flight a block paid one full
    // memory latency per 16-deep K step -- the long split-K weight-gradient GEMMs run one or two blocks per CU, so
    // nothing else hid it.
    auto stage = [&](float4 va, float4 vb, int k0) {
        if (!ok_a(k0)) va = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!ok_b(k0)) vb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (A_UNIT_K) {
            As[(a_k + 0) * GLD + a_r] = va.x;
            As[(a_k + 1) * GLD + a_r] = va.y;
            As[(a_k + 2) * GLD + a_r] = va.z;
            As[(a_k + 3) * GLD + a_r] = va.w;
        } else {
            *reinterpret_cast<float4*>(&As[a_k * GLD + a_r]) = va;
        }
        if (B_UNIT_K) {
            Bs[(b_k + 0) * GLD + b_r] = vb.x;
            Bs[(b_k + 1) * GLD + b_r] = vb.y;
            Bs[(b_k + 2) * GLD + b_r] = vb.z;
            Bs[(b_k + 3) * GLD + b_r] = vb.w;
        } else {
            *reinterpret_cast<float4*>(&Bs[b_k * GLD + b_r]) = vb;
        }
    };
    auto multiply = [&]() {
#pragma unroll
        for (int ks = 0; ks < GK / 4; ++ks) {
            const int kr = 4 * ks + (lane >> 4);
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = As[kr * GLD + 32 * wr + 16 * i + (lane & 15)];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = Bs[kr * GLD + 32 * wc + 16 * j + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };
    float4 va0 = fetch_a(kbeg), vb0 = fetch_b(kbeg);
    float4 va1 = fetch_a(kbeg + GK), vb1 = fetch_b(kbeg + GK);   // past the end: a clamped (re)load that is never staged
    for (int k0 = kbeg; k0 < kend; k0 += 2 * GK) {
        stage(va0, vb0, k0);
        __syncthreads();
        va0 = fetch_a(k0 + 2 * GK);
        vb0 = fetch_b(k0 + 2 * GK);
        __builtin_amdgcn_sched_barrier(0);   // keep the requests in front of the MFMAs (the scheduler sinks them otherwise)
        multiply();
        __syncthreads();
        if (k0 + GK >= kend) break;
        stage(va1, vb1, k0 + GK);
        __syncthreads();
        va1 = fetch_a(k0 + 3 * GK);
        vb1 = fetch_b(k0 + 3 * GK);
        __builtin_amdgcn_sched_barrier(0);
        multiply();
        __syncthreads();
    }
    float* cz = c + (long)blockIdx.z * c_split_stride;
    // Bias and ReLU are applied to all 16 results in registers BEFORE the first store.  A bias load inside the guarded
    // store was 16 dependent round trips per lane; and even with the two values loaded up front, their first use inside
    // each guarded block put an s_waitcnt vmcnt there -- vmcnt also counts stores on gfx9, so every store waited for the
    // previous one to complete.
    if (bias != nullptr) {
        float bv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[j] = bias[min(n0 + 32 * wc + 16 * j + (lane & 15), N - 1)];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += bv[j];
    }
    if (relu) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = fmaxf(acc[i][j][r], 0.0f);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + 32 * wc + 16 * j + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 32 * wr + 16 * i + 4 * (lane >> 4) + r;
                if (m < M && n < N) cz[(long)m * c_ms + n] = acc[i][j][r];
            }
        }
}

// ---------------------------------------------------------------------------------------------------------
// Weights-stationary GEMM for tall row sets (round 4):  out[r][n] = act(sum_k in[r][k] W(k,n) + bias[n]),  rows >> K, N and
// K * N <= 32 K floats -- the LSTM input projection (40 -> 512), the head's first layer (128 -> 256) and its data gradient
// (256 -> 128) over 19456 rows.  The 64 x 64 tiles above re-read W for every row block and run their staging and multiply phases
// one after the other (18.7 / 20.7 / 21.4 us for 0.8 / 1.27 / 1.27 GFLOP).  Here one workgroup per CU keeps its share of W in
// registers for the whole launch (wave w: columns 16 NT w ..; 48-64 VGPRs per lane) and streams 16-row tiles of `in` through a
// double-buffered LDS tile, one barrier per tile: the next tile's 16-byte loads are issued before the MFMAs of the current one.
//   * operands swapped (A = W fragment, B = input rows), so a lane ends up with FOUR CONSECUTIVE columns of one output row:
//     one 16-byte store per 16 x 16 tile and lane instead of four scattered 4-byte ones;
//   * the reduction index is permuted inside a group of 16 (lane (m, q) reads in[m][16 j + 4 q .. + 3] as one ds_read_b128 and
//     feeds component s to MFMA s of the group; the W registers are loaded in the same order), LDS row stride 16 KG + 4 floats:
//     conflict-free 16-byte reads;
//   * two accumulator chains per output tile (even / odd groups) so that a wave with one tile (N = 128) does not issue
//     back-to-back dependent MFMAs.
// KG = groups of 16 along K (K <= 16 KG, K % 4 == 0; columns K .. 16 KG - 1 are zero in LDS and in the registers).
// W_UNIT_K: W(k,n) = w[n * w_stride + k] (a torch Linear weight), else w[k * w_stride + n].
// ---------------------------------------------------------------------------------------------------------
constexpr int RG_THREADS = 512;
// below this many rows the 64 x 64 tiles keep more CUs busy (HOWL_ROWGEMM_MIN_ROWS overrides: the emulator tests run small shapes)
inline int rowgemm_min_rows() {
    const char* e = getenv("HOWL_ROWGEMM_MIN_ROWS");
    return e != nullptr ? atoi(e) : 2048;
}
// NO2 > 0: a thin second layer rides in the epilogue (the classifier head: y2 = out W2^T + b2 with NO2 <= 8 outputs, W2 (NO2, N)
// row-major).  A lane's four columns of an output row are exactly one B-operand value each for four more MFMAs per 16-column tile
// (reduction index permuted as above; A = the wave's slice of W2, rows NO2 .. 15 zero): 4 NT instructions per wave leave the
// wave's share of y2 (16 rows x NO2) in the lanes of groups 0 and 1, the eight waves meet in LDS, and 16 x NO2 results leave per
// tile -- head_out_kernel's launch and its pass over `out` disappear.  (The same products on the vector pipe -- 8 NO2 FMAs, 2 NO2
// cross-lane adds per lane -- cost 4.8 us per launch, almost what the separate kernel took: vector work does not overlap matrix
// work on this part.)
struct RowGemmThin {
    const float* w2;
    const float* b2;
    float* y2;
};
template <int KG, int NT, bool W_UNIT_K, int NO2 = 0>
__global__ __launch_bounds__(RG_THREADS) void rowgemm_kernel(const float* __restrict__ in, RowMap im, const float* __restrict__ w,
                                                             long w_stride, int rows, int K, const float* __restrict__ bias, int relu,
                                                             float* __restrict__ out, long out_stride, RowGemmThin thin) {
    constexpr int LDW = 16 * KG + 4;
    constexpr int NO2P = NO2 > 0 ? NO2 : 1;
    __shared__ __attribute__((aligned(16))) float red2[2][RG_THREADS / 64][16][8];      // [tile parity][wave][row][output]
    constexpr int NL = (64 * KG + RG_THREADS - 1) / RG_THREADS;      // 16-byte pieces of a tile per thread
    __shared__ __attribute__((aligned(16))) float tile[2][16 * LDW + 4];      // + a dump slot for pieces without a place (see stage)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mi = lane & 15, kq = lane >> 4;
    const int nbase = wave * 16 * NT;
    // this wave's share of W, in the order the MFMAs take it
    float wv[NT][KG][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < KG; ++j) {
            const int n = nbase + 16 * i + mi, k0 = 16 * j + 4 * kq;
            // unconditional loads from clamped coordinates, zeroed afterwards (loads under a lane predicate are serialised)
            if (W_UNIT_K) {
                const float4 t = *reinterpret_cast<const float4*>(w + (long)n * w_stride + min(k0, K - 4));
                wv[i][j][0] = k0 < K ? t.x : 0.0f;
                wv[i][j][1] = k0 < K ? t.y : 0.0f;
                wv[i][j][2] = k0 < K ? t.z : 0.0f;
                wv[i][j][3] = k0 < K ? t.w : 0.0f;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = w[(long)min(k0 + e, K - 1) * w_stride + n];
                    wv[i][j][e] = k0 + e < K ? t : 0.0f;
                }
            }
        }
    float4 bv[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
        bv[i] = bias != nullptr ? *reinterpret_cast<const float4*>(bias + nbase + 16 * i + 4 * kq) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 w2v[NT];      // A fragment of the second layer: W2[n2 = mi][nbase + 16 i + 4 kq + e], zero rows from NO2 up
    if constexpr (NO2 > 0) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const float4 t4 = *reinterpret_cast<const float4*>(thin.w2 + (long)min(mi, NO2 - 1) * (RG_THREADS / 64 * 16 * NT) + nbase + 16 * i + 4 * kq);
            w2v[i] = mi < NO2 ? t4 : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    for (int i = tid; i < 2 * (16 * LDW + 4); i += RG_THREADS) (&tile[0][0])[i] = 0.0f;
    // the pieces this thread moves for every tile: piece e = tid + 512 l -> (row e / K4, columns 4 (e % K4) ..)
    const int K4 = K >> 2;
    const int ntiles = (rows + 15) >> 4;
    const float inv_inner = 1.0f / (float)im.inner;
    int prow[NL], pcol[NL], pdst[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const int e = tid + l * RG_THREADS;
        prow[l] = e < 16 * K4 ? e / K4 : -1;
        pcol[l] = e < 16 * K4 ? 4 * (e - (e / K4) * K4) : 0;
        pdst[l] = e < 16 * K4 ? prow[l] * LDW + pcol[l] : 16 * LDW;
    }
    struct Pieces {
        float4 v[NL];
    };
    auto fetch = [&](int t) -> Pieces {
        Pieces pre;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int r = min(16 * min(t, ntiles - 1) + max(prow[l], 0), rows - 1);      // unconditional loads from clamped rows
            // two-level row map without a branch or an integer division: floor(r / inner) through the float reciprocal: the
            // relative error of (r + 0.5) * fl(1 / inner) is <= 1.5 * 2^-23, an absolute 0.19 / inner at the callers' gate of
            // r < 2^20 -- 2.6x inside the 0.5 / inner distance to the nearest integer boundary (a plain stride has
            // inner = 2^30: quotient 0)
            const int q = (int)(((float)r + 0.5f) * inv_inner);
            pre.v[l] = *reinterpret_cast<const float4*>(in + ((long)q * im.s_outer + (long)(r - q * im.inner) * im.s_inner) + pcol[l]);
        }
        return pre;
    };
    // pieces without a place in the tile (narrow K: fewer pieces than threads) go to a dump row behind it, so that no LDS store
    // sits under a lane predicate either
    auto stage = [&](const Pieces& pre, int buf) {
#pragma unroll
        for (int l = 0; l < NL; ++l) *reinterpret_cast<float4*>(&tile[buf][pdst[l]]) = pre.v[l];
    };
    auto multiply = [&](int t, int cur) {
        f32x4 acc[NT][2];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            acc[i][0] = {0.0f, 0.0f, 0.0f, 0.0f};
            acc[i][1] = {0.0f, 0.0f, 0.0f, 0.0f};
        }
        const float* arow = &tile[cur][mi * LDW + 4 * kq];
#pragma unroll
        for (int j = 0; j < KG; ++j) {
            const float4 a = *reinterpret_cast<const float4*>(arow + 16 * j);
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < NT; ++i)
                    acc[i][j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i][j][e], av[e], acc[i][j & 1], 0, 0, 0);
        }
        struct Out {
            float4 v[NT];
        } o;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            float4 v = make_float4(acc[i][0][0] + acc[i][1][0] + bv[i].x, acc[i][0][1] + acc[i][1][1] + bv[i].y,
                                   acc[i][0][2] + acc[i][1][2] + bv[i].z, acc[i][0][3] + acc[i][1][3] + bv[i].w);
            if (relu) v = make_float4(fmaxf(v.x, 0.0f), fmaxf(v.y, 0.0f), fmaxf(v.z, 0.0f), fmaxf(v.w, 0.0f));
            o.v[i] = v;
        }
        if constexpr (NO2 > 0) {
            f32x4 y = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                y = __builtin_amdgcn_mfma_f32_16x16x4f32(w2v[i].x, o.v[i].x, y, 0, 0, 0);
                y = __builtin_amdgcn_mfma_f32_16x16x4f32(w2v[i].y, o.v[i].y, y, 0, 0, 0);
                y = __builtin_amdgcn_mfma_f32_16x16x4f32(w2v[i].z, o.v[i].z, y, 0, 0, 0);
                y = __builtin_amdgcn_mfma_f32_16x16x4f32(w2v[i].w, o.v[i].w, y, 0, 0, 0);
            }
            // D[n2 = 4 kq + r][row = mi]: outputs 0..7 live in lane groups 0 and 1
            if (kq < 2) *reinterpret_cast<float4*>(&red2[cur][wave][mi][4 * kq]) = make_float4(y[0], y[1], y[2], y[3]);
        }
        return o;
    };
    // second layer, after the tile's barrier: every thread folds the eight waves' partials of one (row, output) in a fixed order
    // and stores it -- the 16 x NO2 results of a tile are written by all 512 threads (identical duplicates), because a store under
    // a lane predicate would hide the count of outstanding memory operations from the compiler (see `store`)
    auto store_thin = [&](int t, int cur) {
        if constexpr (NO2 > 0) {
            const int e = tid % (16 * NO2), r = e / NO2, n = e - r * NO2;
            float y = thin.b2[n];
#pragma unroll
            for (int wv_ = 0; wv_ < RG_THREADS / 64; ++wv_) y += red2[cur][wv_][r][n];
            thin.y2[(long)min(16 * t + r, rows - 1) * NO2 + n] = y;
        }
    };
    // D[n_local = 4 kq + r][m = mi]: four consecutive columns of output row 16 t + mi.  Rows past the end were loaded from the
    // last row (fetch clamps), so their results ARE the last row's, bit for bit: they are stored there again, and no store sits
    // under a lane predicate -- which would hide the number of outstanding memory operations from the compiler and turn the
    // wait in front of the staging stores into vmcnt(0).
    auto store = [&](const auto& o, int t) {
        const int row = min(16 * t + mi, rows - 1);
#pragma unroll
        for (int i = 0; i < NT; ++i) *reinterpret_cast<float4*>(out + (long)row * out_stride + nbase + 16 * i + 4 * kq) = o.v[i];
    };
    // Two tiles are on their way while one is multiplied: tile t + 2 G is requested at the top of trip t; tile t + G (requested a
    // trip earlier) goes to the other LDS buffer after the MFMAs and BEFORE this trip's result stores, so the wait for its data
    // covers loads only (the counter is in order).  The loop is unrolled by two: the two register sets swap roles without copies.
    const int G = gridDim.x;
    int t = blockIdx.x;
    __syncthreads();          // zero fill before the first pieces land
    Pieces p0 = fetch(t);
    Pieces p1 = fetch(t + G);
    stage(p0, 0);
    __syncthreads();
    for (; t < ntiles; t += 2 * G) {
        p0 = fetch(t + 2 * G);
        __builtin_amdgcn_sched_barrier(0);      // keep the requests in front of the MFMAs
        const auto o0 = multiply(t, 0);
        stage(p1, 1);
        __builtin_amdgcn_sched_barrier(0);
        store(o0, t);
        __syncthreads();      // next tile complete; every wave is past its reads of the current one
        store_thin(t, 0);
        if (t + G >= ntiles) break;
        p1 = fetch(t + 3 * G);
        __builtin_amdgcn_sched_barrier(0);
        const auto o1 = multiply(t + G, 1);
        stage(p0, 0);
        __builtin_amdgcn_sched_barrier(0);
        store(o1, t + G);
        __syncthreads();
        store_thin(t + G, 1);
    }
}

// deterministic sum of `nparts` slabs of n floats (split-K partials): a block owns 64 outputs, its 4 waves take the
// slabs g = wave, wave+4, ... (four loads in flight each) and combine through LDS in a fixed order
__global__ __launch_bounds__(256) void sum_slabs_kernel(const float* __restrict__ part, int nparts, long n,
                                                        float* __restrict__ out) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + lane;
    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (i < n) {
        int g = rg;
        for (; g + 12 < nparts; g += 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] += part[(long)(g + 4 * u) * n + i];
        }
        for (; g < nparts; g += 4) s[0] += part[(long)g * n + i];
    }
    red[rg][lane] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (rg == 0 && i < n) out[i] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

// Several slab sums in ONE launch: the split-K weight gradients and bias column sums of a backward call each end in a
// sum_slabs of their own scratch region; queued here (SlabSums::add) and folded together, a call pays one ~5 us launch
// instead of three to five.  blockIdx.y picks the job, blocks beyond a job's 64-column count return.
struct SlabJob {
    const float* part;
    float* out;
    float* out2;     // optional second destination (the LSTM's two bias vectors share one gradient)
    long n;
    int nparts;
    long stride;     // distance between slabs in floats (= n when the slabs are packed)
};
constexpr int MAX_SLAB_JOBS = 8;
struct SlabJobs {
    SlabJob j[MAX_SLAB_JOBS];
};
// Optional optimiser step on the folded gradients (round 5: the seq-lstm step's AdamW launch -- 5 us of a 265-us step -- and its
// pass over the gradients): p / m / v are the flat parameter and moment buffers, g0 the flat gradient buffer every job's `out`
// points into; element `out - g0 + i` of the three is updated with the sum just written to out[i].
struct SlabAdamW {
    float* p;
    const float* g0;
    float* m;
    float* v;
    HowlAdamWCoef c;
    int on;
};
// Optional rider of the fold launch (round 6): the batch mean of a CTC loss, loss = mean_b nll_b / max(L_b, 1) (CTCLoss reduction
// "mean", ctc_mean_kernel's summation order over 256 threads) -- one more block row instead of a launch of its own.
struct SlabMean {
    const float* nll;
    const long long* target_lengths;
    int B;
    float* loss;
};
__device__ __forceinline__ void ctc_mean_256(const float* __restrict__ nll, const long long* __restrict__ target_lengths, int B,
                                             float* __restrict__ loss) {
    __shared__ double mred[4];
    double acc = 0.0;
    for (int b = threadIdx.x; b < B && threadIdx.x < 256; b += 256) {      // (threads past 256 of a wider block: idle)
        const long long L = target_lengths[b];
        acc += (double)(nll[b] / (float)(L > 0 ? L : 1));
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0 && threadIdx.x < 256) mred[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (float)((((mred[0] + mred[1]) + mred[2]) + mred[3]) / (double)B);
}
__global__ __launch_bounds__(256) void sum_slabs_multi_kernel(SlabJobs jobs, SlabAdamW opt, SlabMean mean, int njobs) {
    if ((int)blockIdx.y == njobs) {      // the rider's block row
        if (blockIdx.x == 0) ctc_mean_256(mean.nll, mean.target_lengths, mean.B, mean.loss);
        return;
    }
    __shared__ float red[4][64];
    const SlabJob& jb = jobs.j[blockIdx.y];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + lane;
    if ((long)blockIdx.x * 64 >= jb.n) return;
    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (i < jb.n) {
        // eight loads in flight per lane: the long jobs (128 .. 512 slabs) are a chain of dependent round trips otherwise
        int g = rg;
        for (; g + 28 < jb.nparts; g += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = jb.part[(long)(g + 4 * u) * jb.stride + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u & 3] += v[u];
        }
        for (; g + 12 < jb.nparts; g += 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] += jb.part[(long)(g + 4 * u) * jb.stride + i];
        }
        for (; g < jb.nparts; g += 4) s[0] += jb.part[(long)g * jb.stride + i];
    }
    red[rg][lane] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (rg == 0 && i < jb.n) {
        const float t = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
        jb.out[i] = t;
        if (jb.out2 != nullptr) jb.out2[i] = t;
        if (opt.on) {
            howl_adamw_element(opt.p, opt.m, opt.v, (size_t)(jb.out - opt.g0) + (size_t)i, t, opt.c);
            if (jb.out2 != nullptr) howl_adamw_element(opt.p, opt.m, opt.v, (size_t)(jb.out2 - opt.g0) + (size_t)i, t, opt.c);
        }
    }
}

// collector: pass to wgrad_gemm / colsum to defer their final sum, call flush() once per backward call
struct SlabSums {
    SlabJobs jobs;
    int count = 0;
    long max_n = 0;
    bool overflow = false;     // a caller queued more than MAX_SLAB_JOBS folds without a flush: reported by flush()
    SlabMean mean{nullptr, nullptr, 0, nullptr};      // optional rider of the next flush (ctc_mean_256)
    void add(const float* part, int nparts, long n, float* out, float* out2 = nullptr) { add_strided(part, nparts, n, n, out, out2); }
    // slabs that sit `stride` floats apart (several partial results interleaved per producer block)
    void add_strided(const float* part, int nparts, long stride, long n, float* out, float* out2 = nullptr) {
        if (count >= MAX_SLAB_JOBS) {           // never write past the kernel-argument array
            overflow = true;
            return;
        }
        jobs.j[count++] = SlabJob{part, out, out2, n, nparts, stride};
        max_n = n > max_n ? n : max_n;
    }
    // true when the queued folds write every element of the flat gradient buffer [g0, g0 + n) exactly once: only then may the
    // optimiser step ride in the fold (a gradient that reaches the buffer by another path would miss its update)
    bool covers(const float* g0, size_t n) const {
        if (overflow || count == 0) return false;
        size_t total = 0;
        for (int q = 0; q < count; ++q) {
            const SlabJob& jb = jobs.j[q];
            for (const float* o : {(const float*)jb.out, (const float*)jb.out2}) {
                if (o == nullptr) continue;
                if (o < g0 || o + jb.n > g0 + n) return false;
                total += (size_t)jb.n;
            }
        }
        return total == n;      // (ranges of distinct parameters do not overlap: equal totals = a partition)
    }
    // returns false (and sets the library's error string) when a fold was dropped by add(): the caller fails its call
    bool flush(hipStream_t s, const SlabAdamW* opt = nullptr) {
        if (overflow) {
            howl_set_error("SlabSums: more than %d folds queued between two flushes", MAX_SLAB_JOBS);
            overflow = false;
            count = 0;
            return false;
        }
        if (count == 0 && mean.nll == nullptr) return true;
        const int rider = mean.nll != nullptr ? 1 : 0;
        hipLaunchKernelGGL(sum_slabs_multi_kernel, dim3((unsigned)std::max<long>((max_n + 63) / 64, 1), count + rider), dim3(256), 0, s, jobs,
                           opt != nullptr ? *opt : SlabAdamW{nullptr, nullptr, nullptr, nullptr, HowlAdamWCoef{}, 0}, mean, count);
        count = 0;
        max_n = 0;
        mean = SlabMean{nullptr, nullptr, 0, nullptr};
        return true;
    }
};

// column sums of a (rows, n) matrix with a row map, two stages so that long row counts use the whole chip:
// block (x = 64 columns, y = row chunk) writes part[y][col] in fp32 from an fp64 running sum; sum_slabs_kernel folds
// the chunks in a fixed order (deterministic).
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, RowMap rm, int rows, int n,
                                                     int rows_per_chunk, float* __restrict__ part) {
    __shared__ double red[4][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * rows_per_chunk;
    const int r1 = min(rows, r0 + rows_per_chunk);
    double s = 0.0;
    if (col < n) {
        double s4[4] = {0.0, 0.0, 0.0, 0.0};   // four independent loads in flight (a dependent one-load loop is latency-bound)
        int r = r0 + rg;
        for (; r + 12 < r1; r += 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s4[u] += (double)x[rmap(rm, r + 4 * u) + col];
        }
        for (; r < r1; r += 4) s4[0] += (double)x[rmap(rm, r) + col];
        s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
    red[rg][lane] = s;
    __syncthreads();
    if (rg == 0 && col < n)
        part[(size_t)blockIdx.y * n + col] = (float)(((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane]);
}

// dz = dy * (y > 0), elementwise (ReLU backward of the head's hidden layer); dz may be dy (in place)
__global__ void relu_bwd_kernel(const float* dy, const float* __restrict__ y, long n, float* dz) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dz[i] = y[i] > 0.0f ? dy[i] : 0.0f;
}


constexpr int BIG = 1 << 30;
inline RowMap lin(long stride) { return RowMap{BIG, 0, stride}; }
inline bool is_lin(const RowMap& r) { return r.inner == BIG; }

// `thin` (optional): RowGemmThin of a second layer with thin_out <= 8 outputs for the weights-stationary kernel; *thin_done tells
// the caller whether it rode along (otherwise the caller launches the second layer itself)
int gemm(hipStream_t s, bool a_major_k, const float* a, RowMap am, long a_ks, RowMap ak, const float* b, RowMap bk, long b_ns,
         int M, int N, int K, int splits, const float* bias, int relu, float* c, long c_ms, long c_split_stride,
         const RowGemmThin* thin = nullptr, int thin_out = 0, bool* thin_done = nullptr) {
    if (thin_done != nullptr) *thin_done = false;
    const int kps = ((K + splits - 1) / splits + GK - 1) / GK * GK;
    dim3 grid((N + GT - 1) / GT, (M + GT - 1) / GT, (K + kps - 1) / kps);
    HowlProfScope prof("gemm", s, 2.0 * (double)M * N * K);
    // tall row sets against a small weight matrix: the weights-stationary kernel
    {
        const bool w_unit_k = is_lin(bk) && bk.s_inner == 1;      // W(k,n) = b[n * b_ns + k]
        const bool w_unit_n = b_ns == 1 && is_lin(bk);            // W(k,n) = b[k * bk.s_inner + n]
        const long w_stride = w_unit_k ? b_ns : bk.s_inner;
        auto al16 = [](const void* p_) { return (reinterpret_cast<uintptr_t>(p_) & 15) == 0; };
        const bool fits = a_major_k && a_ks == 1 && splits == 1 && M < (1 << 20) && (w_unit_k || w_unit_n) && M >= rowgemm_min_rows() && (K & 3) == 0 &&
                          (am.s_outer & 3) == 0 && (am.s_inner & 3) == 0 && al16(a) && al16(b) && al16(c) && (c_ms & 3) == 0 &&
                          (!w_unit_k || (w_stride & 3) == 0) && (bias == nullptr || al16(bias)) && getenv("HOWL_GEMM_NO_ROWGEMM") == nullptr;
        if (fits) {
            const int ntiles = (M + 15) / 16;
            const int blocks = std::min(ntiles, howl_num_cus());
#define HOWL_ROWGEMM(KG_, NT_)                                                                                                \
    do {                                                                                                                      \
        if (w_unit_k)                                                                                                         \
            hipLaunchKernelGGL((rowgemm_kernel<KG_, NT_, true>), dim3(blocks), dim3(RG_THREADS), 0, s, a, am, b, w_stride, M, K, \
                               bias, relu, c, c_ms, RowGemmThin{});                                                           \
        else                                                                                                                  \
            hipLaunchKernelGGL((rowgemm_kernel<KG_, NT_, false>), dim3(blocks), dim3(RG_THREADS), 0, s, a, am, b, w_stride, M, K, \
                               bias, relu, c, c_ms, RowGemmThin{});                                                           \
        return 1;                                                                                                             \
    } while (0)
            if (N == 512 && K <= 48) HOWL_ROWGEMM(3, 4);
            if (N == 256 && K > 64 && K <= 128 && w_unit_k && thin != nullptr && thin_out >= 1 && thin_out <= 8 && c_ms == N &&
                al16(thin->w2)) {
                // the head: first layer + thin second layer in one launch
#define HOWL_ROWGEMM_THIN(NO_)                                                                                                \
    case NO_:                                                                                                                 \
        hipLaunchKernelGGL((rowgemm_kernel<8, 2, true, NO_>), dim3(blocks), dim3(RG_THREADS), 0, s, a, am, b, w_stride, M, K,  \
                           bias, relu, c, c_ms, *thin);                                                                       \
        break;
                switch (thin_out) {
                    HOWL_ROWGEMM_THIN(1) HOWL_ROWGEMM_THIN(2) HOWL_ROWGEMM_THIN(3) HOWL_ROWGEMM_THIN(4) HOWL_ROWGEMM_THIN(5)
                    HOWL_ROWGEMM_THIN(6) HOWL_ROWGEMM_THIN(7) HOWL_ROWGEMM_THIN(8)
                }
#undef HOWL_ROWGEMM_THIN
                *thin_done = true;
                return 1;
            }
            if (N == 256 && K > 64 && K <= 128) HOWL_ROWGEMM(8, 2);
            if (N == 128 && K > 128 && K <= 256) HOWL_ROWGEMM(16, 1);
#undef HOWL_ROWGEMM
        }
    }
    // operands whose unit-stride runs are 16-byte aligned multiples of 4 floats take the vector kernel
    {
        auto map_ok = [](const RowMap& r) { return (r.s_outer & 3) == 0 && (r.s_inner & 3) == 0; };
        auto aligned = [](const float* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        // A: unit stride along k (a_major_k, rows mapped by am) or along m (am == lin(1), k mapped by ak)
        const bool a_fits = aligned(a) && (a_major_k ? (a_ks == 1 && map_ok(am) && (K & 3) == 0)
                                                     : (is_lin(am) && am.s_inner == 1 && map_ok(ak) && (M & 3) == 0));
        // B: unit stride along k (bk == lin(1), column stride b_ns) or along n (b_ns == 1, k mapped by bk)
        const bool b_unit_k = is_lin(bk) && bk.s_inner == 1 && b_ns != 1;
        const bool b_unit_n = b_ns == 1;
        const bool b_fits = aligned(b) && (b_unit_k ? ((b_ns & 3) == 0 && (K & 3) == 0) : (b_unit_n && map_ok(bk) && (N & 3) == 0));
        if (a_fits && b_fits) {
            const RowMap amap = a_major_k ? am : ak;
            const RowMap bmap = b_unit_k ? lin(b_ns) : bk;
            // the maps that are indexed by k inside the kernel (the other ones are evaluated once per thread)
            const bool klin = (a_major_k || is_lin(amap)) && (b_unit_k || is_lin(bmap));
#define HOWL_GEMM_VEC(AK, BK)                                                                                                  \
    do {                                                                                                                       \
        if (klin)                                                                                                              \
            hipLaunchKernelGGL((gemm_vec_kernel<AK, BK, true>), grid, dim3(256), 0, s, a, amap, b, bmap, M, N, K, kps, bias,   \
                               relu, c, c_ms, c_split_stride);                                                                 \
        else                                                                                                                   \
            hipLaunchKernelGGL((gemm_vec_kernel<AK, BK, false>), grid, dim3(256), 0, s, a, amap, b, bmap, M, N, K, kps, bias,  \
                               relu, c, c_ms, c_split_stride);                                                                 \
    } while (0)
            if (a_major_k && b_unit_k) HOWL_GEMM_VEC(true, true);
            else if (a_major_k) HOWL_GEMM_VEC(true, false);
            else if (b_unit_k) HOWL_GEMM_VEC(false, true);
            else HOWL_GEMM_VEC(false, false);
#undef HOWL_GEMM_VEC
            return (int)grid.z;
        }
    }
    if (a_major_k)
        hipLaunchKernelGGL(gemm_kernel<true>, grid, dim3(256), 0, s, a, am, a_ks, ak, b, bk, b_ns, M, N, K, kps, bias, relu, c,
                           c_ms, c_split_stride);
    else
        hipLaunchKernelGGL(gemm_kernel<false>, grid, dim3(256), 0, s, a, am, a_ks, ak, b, bk, b_ns, M, N, K, kps, bias, relu,
                           c, c_ms, c_split_stride);
    return (int)grid.z;
}

// Weight gradient with a THIN output (N_out <= 8: the 5-label Linear head, MobileNet's 3-channel downsample conv): a 64x64
// MFMA tile would be > 87 % padding and the operands do not meet the vector path's alignment, so this is a plain streaming
// reduction instead: lane -> input column k (narrow matrices pack 64/Kp rows per wave, Kp = K_in rounded up to a power of
// two <= 32), the N_out values of a row are broadcast loads, four rows in flight per lane, fp32 partials per row chunk folded
// by sum_slabs_kernel in a fixed order.   part[chunk][n][k]
template <int NO>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(const float* __restrict__ dout, RowMap dm, const float* __restrict__ in,
                                                         RowMap im, int k_in, int kp, int rows, int rows_per_chunk,
                                                         float* __restrict__ part) {
    __shared__ float red[4][NO][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int rsub = 64 / kp;                       // rows per wave iteration
    const int kl = lane & (kp - 1), rs = lane / kp;
    const int k = blockIdx.x * 64 + kl;
    const int r0 = blockIdx.y * rows_per_chunk;
    const int r1 = min(rows, r0 + rows_per_chunk);
    float acc[NO];
#pragma unroll
    for (int n = 0; n < NO; ++n) acc[n] = 0.0f;
    if (k < k_in) {
        const int step = 4 * rsub;
        for (int r = r0 + rg * rsub + rs; r < r1; r += 4 * step) {
            float b[4], a[4][NO];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = r + u * step;
                const bool ok = rr < r1;
                const int rc = ok ? rr : r;       // clamped: branch-free loads, all in flight
                const float bv = in[rmap(im, rc) + k];
                b[u] = ok ? bv : 0.0f;
                const float* ap = dout + rmap(dm, rc);
#pragma unroll
                for (int n = 0; n < NO; ++n) a[u][n] = ap[n];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int n = 0; n < NO; ++n) acc[n] = fmaf(a[u][n], b[u], acc[n]);
        }
    }
#pragma unroll
    for (int n = 0; n < NO; ++n) red[rg][n][lane] = acc[n];
    __syncthreads();
    if (rg == 0 && rs == 0 && k < k_in) {
#pragma unroll
        for (int n = 0; n < NO; ++n) {
            float t = 0.0f;
            for (int w = 0; w < 4; ++w)
                for (int q = 0; q < rsub; ++q) t += red[w][n][q * kp + kl];
            part[((size_t)blockIdx.y * NO + n) * k_in + k] = t;
        }
    }
}

// Weight gradient of a WIDE layer over many rows (the LSTM's W_hh / W_ih and the head's first layer: 128..512 outputs, 19,456 rows):
//   part[z][m][n] = sum over the rows k of split z of dout[dm(k) + m] * in[im(k) + n]
// Both operands are row-major with the reduction index as the ROW, so a K tile of sixteen rows goes to LDS as it is ([k][column],
// 16-byte stores, no transpose) and a lane's fragment values are plain 4-byte reads.  128 (m) x TN (n) block tile, sixteen waves
// (4 x 4, 32 x TN/4 each: four waves per SIMD, so that fragment reads, staging and address arithmetic of some run under the
// MFMAs of others), two LDS buffers and two K tiles in flight in registers: one barrier per K tile, the next tile's stores
// land while the other buffer is multiplied.  The launcher sizes the split count so that the grid is ONE block per CU (or two):
// the 64 x 64 kernel moved every operand element through the cache hierarchy 2-8 times (160 MB for 50 MB of operands in the
// W_hh gradient) and its 1216 blocks did not overlap their staging with their MFMAs; here each element is read once or twice.
constexpr int WG_M = 128, WG_K = 16, WG_THREADS = 1024;
constexpr int WG_LDA = WG_M + 16;                   // pitch = 16 mod 32 banks: the two k rows of a 32-lane half of a fragment read do not collide (with + 4 they did: -1 %)
constexpr int WG_LDB = 192 + 16;                    // the widest B tile (the dual job's 128 + 64 columns), same pitch rule
// A second input matrix for the SAME dout rows (round 5: the LSTM's dW_hh = dG^T H and dW_ih = dG^T X as one job): its N2 <= 64
// columns sit behind the first input's 128 in the block's B tile (TN = 192), so dG is read -- and staged, and its fragments
// fetched -- once for both products; part2 receives the [splits][M][N2] slabs.
struct WgradSecond {
    const float* in2;
    RowMap im2;
    int N2;
    float* part2;
};
// (bx, by, bz) = the block's (input-column tile, output-column tile, row split); As: [2][WG_K * WG_LDA], Bs: [2][WG_K * WG_LDB]
// floats of LDS.  DUAL: TN = 192, N = 128 (the first input fills columns 0..127 exactly), KMAP_LIN = false.
template <int TN, bool KMAP_LIN, bool DUAL = false>
__device__ __forceinline__ void wgrad_big_body(const float* __restrict__ dout, RowMap dm, const float* __restrict__ in, RowMap im,
                                               int M, int N, int K, int k_per_split, float* __restrict__ part, int bx, int by,
                                               int bz, float (*As)[WG_K * WG_LDA], float (*Bs)[WG_K * WG_LDB],
                                               const WgradSecond sec = WgradSecond{nullptr, RowMap{1, 0, 0}, 0, nullptr}) {
    static_assert(!DUAL || (TN == 192 && !KMAP_LIN), "dual job: 128 + 64 columns, general row maps");
    constexpr int LDA = WG_LDA, LDB = TN + 16;
    constexpr int NJ = TN / 64;                     // 16-column tiles per wave (sixteen waves: 4 x 4, 32 x TN/4 each)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int m0 = by * WG_M, n0 = bx * TN;
    const int kbeg = bz * k_per_split, kend = min(K, kbeg + k_per_split);
    // One 16-byte piece of ONE operand per thread and K tile (wave-uniform role): threads 0..511 carry dout's tile
    // (k = t / 32, column 4 (t % 32)), threads 512.. the input's (TN = 128: the same map; TN = 64: k = t / 16, 256 threads).
    const bool is_a = tid < 512;
    const int t2 = tid & 511;
    const int p_k = (is_a || TN != 64) ? t2 >> 5 : (t2 >> 4) & 15;
    const int p_c = (is_a || TN != 64) ? (t2 & 31) * 4 : (t2 & 15) * 4;
    const bool p_thread = is_a || TN != 64 || t2 < 256;
    const float* src = is_a ? dout : in;
    const RowMap rm = is_a ? dm : im;
    const int col = is_a ? min(m0 + p_c, M - 4) : min(n0 + p_c, N - 4);      // clamped loads, zeroed when staged
    const bool p_ok = p_thread && (is_a ? m0 + p_c < M : n0 + p_c < N);
    float* const dst0 = is_a ? &As[0][p_k * LDA + p_c] : &Bs[0][p_k * LDB + p_c];
    float* const dst1 = is_a ? &As[1][p_k * LDA + p_c] : &Bs[1][p_k * LDB + p_c];
    // Row cursor: successive fetches are sixteen rows apart, so the two-level row maps ((b, t) rows of buffers with a gap per
    // utterance) advance by addition -- one division at the start instead of one per K tile (~50 VALU instructions each).  Past
    // the end of the split the cursor stays on the split's last row (the staged value is zeroed anyway).
    int ck = min(kbeg + p_k, kend - 1);
    int cin = KMAP_LIN ? 0 : ck % rm.inner;
    long coff = KMAP_LIN ? (long)ck * rm.s_inner : (long)(ck / rm.inner) * rm.s_outer + (long)cin * rm.s_inner;
    auto fetch = [&]() {     // loads the cursor's row and moves on: calls are in K-tile order (kbeg, kbeg + 16, ...)
        const float4 v = *reinterpret_cast<const float4*>(src + coff + col);
        const int step = min(WG_K, kend - 1 - ck);
        ck += step;
        coff += (long)step * rm.s_inner;
        if (!KMAP_LIN) {
            cin += step;
            while (cin >= rm.inner) {
                cin -= rm.inner;
                coff += rm.s_outer - (long)rm.inner * rm.s_inner;
            }
        }
        return v;
    };
    auto stage = [&](float* dst, float4 v, int k0) {
        if (!(p_ok && k0 + p_k < kend)) v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p_thread) *reinterpret_cast<float4*>(dst) = v;
    };
    // DUAL: the second input's 16 x N2 tile = 4 N2 pieces, one more per K tile for the first 4 N2 threads (row e / (N2 / 4),
    // columns 4 (e % (N2 / 4)) ..), with a row cursor of its own; B columns 128 + N2 .. 191 stay zero from the start
    const int q4 = DUAL ? sec.N2 >> 2 : 1;
    const bool x_thread = DUAL && tid < 16 * q4;
    const int x_k = x_thread ? tid / q4 : 0;
    const int x_c = x_thread ? 4 * (tid - x_k * q4) : 0;
    int xk = min(kbeg + x_k, kend - 1);
    int xin = DUAL ? xk % sec.im2.inner : 0;
    long xoff = DUAL ? (long)(xk / sec.im2.inner) * sec.im2.s_outer + (long)xin * sec.im2.s_inner : 0;
    auto fetch_x = [&]() {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (DUAL) {
            if (x_thread) v = *reinterpret_cast<const float4*>(sec.in2 + xoff + x_c);
            const int step = min(WG_K, kend - 1 - xk);
            xk += step;
            xoff += (long)step * sec.im2.s_inner;
            xin += step;
            while (xin >= sec.im2.inner) {
                xin -= sec.im2.inner;
                xoff += sec.im2.s_outer - (long)sec.im2.inner * sec.im2.s_inner;
            }
        }
        return v;
    };
    auto stage_x = [&](int buf, float4 v, int k0) {
        if constexpr (DUAL) {
            if (!(k0 + x_k < kend)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (x_thread) *reinterpret_cast<float4*>(&Bs[buf][x_k * LDB + 128 + x_c]) = v;
        }
    };
    if constexpr (DUAL) {
        for (int i = tid; i < 2 * WG_K * WG_LDB; i += WG_THREADS) (&Bs[0][0])[i] = 0.0f;
        __syncthreads();
    }
    f32x4 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = {0.0f, 0.0f, 0.0f, 0.0f};
    auto multiply = [&](int buf) {
        const float* as = &As[buf][(lane >> 4) * LDA + 32 * wr + (lane & 15)];
        const float* bs = &Bs[buf][(lane >> 4) * LDB + (TN / 4) * wc + (lane & 15)];
#pragma unroll
        for (int ks = 0; ks < WG_K / 4; ++ks) {
            float af[2], bf[NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = as[4 * ks * LDA + 16 * i];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bf[j] = bs[4 * ks * LDB + 16 * j];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };
    // (Four tiles in flight instead of two -- four per trip -- measured 71 us against 64-66 for the three jobs of the seq-lstm step:
    // the loop is not waiting for its loads, and the all-zero tiles that round the count up are paid in full.)
    // (Two K tiles per BARRIER -- four LDS buffers, 32 MFMAs per wave between two barriers instead of 16 -- measured 68.6 us against
    // 64-66 as well: neither the load latency nor the barrier count is what keeps the matrix pipe at 0.485 here.)
    // Two K tiles per trip, no early exit (an odd tile count multiplies one all-zero tile at the end): with `break`s in the body
    // the compiler's vmcnt bookkeeping gives up and every staging step waits for ALL outstanding loads (vmcnt(0)).
    float4 v0 = fetch(), x0 = fetch_x();
    float4 v1 = fetch(), x1 = fetch_x();      // past the end: clamped re-loads, staged as zeros
    stage(dst0, v0, kbeg);
    stage_x(0, x0, kbeg);
    __syncthreads();
    v0 = fetch();
    x0 = fetch_x();
    const int pairs = ((kend - kbeg + WG_K - 1) / WG_K + 1) / 2;
    int k0 = kbeg;
    for (int p = 0; p < pairs; ++p, k0 += 2 * WG_K) {
        __builtin_amdgcn_sched_barrier(0);   // keep the requests in front of the MFMAs
        multiply(0);
        stage(dst1, v1, k0 + WG_K);
        stage_x(1, x1, k0 + WG_K);
        __syncthreads();
        v1 = fetch();
        x1 = fetch_x();
        __builtin_amdgcn_sched_barrier(0);
        multiply(1);
        stage(dst0, v0, k0 + 2 * WG_K);
        stage_x(0, x0, k0 + 2 * WG_K);
        __syncthreads();
        v0 = fetch();
        x0 = fetch_x();
    }
    float* pz = part + (long)bz * M * N;
    float* pz2 = DUAL ? sec.part2 + (long)bz * M * sec.N2 : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + (TN / 4) * wc + 16 * j + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 32 * wr + 16 * i + 4 * (lane >> 4) + r;
                if (m < M && n < N) pz[(long)m * N + n] = acc[i][j][r];
                if constexpr (DUAL) {
                    if (m < M && n >= 128 && n - 128 < sec.N2) pz2[(long)m * sec.N2 + n - 128] = acc[i][j][r];
                }
            }
        }
}

template <int TN, bool KMAP_LIN>
__global__ __launch_bounds__(WG_THREADS) void wgrad_big_kernel(const float* __restrict__ dout, RowMap dm,
                                                              const float* __restrict__ in, RowMap im, int M, int N, int K,
                                                              int k_per_split, float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float As[2][WG_K * WG_LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][WG_K * WG_LDB];
    wgrad_big_body<TN, KMAP_LIN>(dout, dm, in, im, M, N, K, k_per_split, part, blockIdx.x, blockIdx.y, blockIdx.z, As, Bs);
}

// Several weight gradients in ONE launch (round 4: the seq-lstm step's W_hh, W_ih and head layer-1 gradients -- three launches of
// one block per CU each, whose ramps, prologues and tails (4-5 us apiece) did not overlap): blocks are numbered job by job, the
// job with the longest blocks first, so that a CU picks up a block of the next job the moment its current one retires.
struct WgradJob {
    const float* dout;
    RowMap dm;
    const float* in;
    RowMap im;
    int M, N, K, kps;
    float* part;
    int gx, gy, gz;
    int tn, klin;
    WgradSecond sec;      // tn == 192: the dual job's second input
};
constexpr int MAX_WGRAD_JOBS = 4;
struct WgradJobs {
    WgradJob j[MAX_WGRAD_JOBS];
    int count = 0;
    double flops = 0.0;
};
__global__ __launch_bounds__(WG_THREADS) void wgrad_big_multi_kernel(WgradJobs jobs) {
    __shared__ __attribute__((aligned(16))) float As[2][WG_K * WG_LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][WG_K * WG_LDB];
    int b = blockIdx.x, q = 0;
    while (q + 1 < jobs.count && b >= jobs.j[q].gx * jobs.j[q].gy * jobs.j[q].gz) {
        b -= jobs.j[q].gx * jobs.j[q].gy * jobs.j[q].gz;
        ++q;
    }
    const WgradJob& jb = jobs.j[q];
    const int bx = b % jb.gx, by = (b / jb.gx) % jb.gy, bz = b / (jb.gx * jb.gy);
#define HOWL_WG_BODY(TN_, KL_) \
    wgrad_big_body<TN_, KL_>(jb.dout, jb.dm, jb.in, jb.im, jb.M, jb.N, jb.K, jb.kps, jb.part, bx, by, bz, As, Bs)
    if (jb.tn == 192) {
        wgrad_big_body<192, false, true>(jb.dout, jb.dm, jb.in, jb.im, jb.M, jb.N, jb.K, jb.kps, jb.part, bx, by, bz, As, Bs, jb.sec);
    } else if (jb.tn == 128) {
        if (jb.klin) HOWL_WG_BODY(128, true);
        else HOWL_WG_BODY(128, false);
    } else {
        if (jb.klin) HOWL_WG_BODY(64, true);
        else HOWL_WG_BODY(64, false);
    }
#undef HOWL_WG_BODY
}
// An EIGHT-wave (512-thread) form of wgrad_big_body<128, false> for one block of a collected job, for blocks that ride in the
// launch of a 512-thread kernel which leaves CUs idle (round 5: the head's first-layer weight gradient inside the backward
// recurrence's launch -- on a second queue the same overlap cost two cross-queue waits of ~7 us each on the critical path).  Same
// tile (128 x 128), same K tiles and split, same order of the products of every output element -- bit-identical slabs -- with
// the sixteen-wave body's 4 x 4 waves of 32 x 32 as 2 x 4 waves of 64 x 32, and one 16-byte piece of EACH operand per thread and
// K tile.  Two waves per SIMD instead of four: slower per block, which is what idle CUs are for.
__device__ __forceinline__ void wgrad_w8_body(const WgradJob& jb, int bidx, float (*As)[WG_K * WG_LDA], float (*Bs)[WG_K * WG_LDA]) {
    constexpr int LDA = WG_LDA, LDB = WG_LDA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int bx = bidx % jb.gx, by = (bidx / jb.gx) % jb.gy, bz = bidx / (jb.gx * jb.gy);
    const int M = jb.M, N = jb.N, K = jb.K;
    const int m0 = by * WG_M, n0 = bx * 128;
    const int kbeg = bz * jb.kps, kend = min(K, kbeg + jb.kps);
    const int p_k = tid >> 5, p_c = (tid & 31) * 4;
    const int cola = min(m0 + p_c, M - 4), colb = min(n0 + p_c, N - 4);      // clamped loads, zeroed when staged
    const bool oka = m0 + p_c < M, okb = n0 + p_c < N;
    int ck = min(kbeg + p_k, kend - 1);
    int cina = ck % jb.dm.inner, cinb = ck % jb.im.inner;
    long coffa = (long)(ck / jb.dm.inner) * jb.dm.s_outer + (long)cina * jb.dm.s_inner;
    long coffb = (long)(ck / jb.im.inner) * jb.im.s_outer + (long)cinb * jb.im.s_inner;
    struct Pair {
        float4 a, b;
    };
    auto fetch = [&]() {     // loads the cursors' rows and moves on: calls are in K-tile order (kbeg, kbeg + 16, ...)
        Pair v;
        v.a = *reinterpret_cast<const float4*>(jb.dout + coffa + cola);
        v.b = *reinterpret_cast<const float4*>(jb.in + coffb + colb);
        const int step = min(WG_K, kend - 1 - ck);
        ck += step;
        coffa += (long)step * jb.dm.s_inner;
        coffb += (long)step * jb.im.s_inner;
        cina += step;
        cinb += step;
        while (cina >= jb.dm.inner) {
            cina -= jb.dm.inner;
            coffa += jb.dm.s_outer - (long)jb.dm.inner * jb.dm.s_inner;
        }
        while (cinb >= jb.im.inner) {
            cinb -= jb.im.inner;
            coffb += jb.im.s_outer - (long)jb.im.inner * jb.im.s_inner;
        }
        return v;
    };
    auto stage = [&](int buf, Pair v, int k0) {
        const bool live = k0 + p_k < kend;
        if (!(oka && live)) v.a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!(okb && live)) v.b = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(&As[buf][p_k * LDA + p_c]) = v.a;
        *reinterpret_cast<float4*>(&Bs[buf][p_k * LDB + p_c]) = v.b;
    };
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = {0.0f, 0.0f, 0.0f, 0.0f};
    auto multiply = [&](int buf) {
        const float* as = &As[buf][(lane >> 4) * LDA + 64 * wr + (lane & 15)];
        const float* bs = &Bs[buf][(lane >> 4) * LDB + 32 * wc + (lane & 15)];
#pragma unroll
        for (int ks = 0; ks < WG_K / 4; ++ks) {
            float af[4], bf[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = as[4 * ks * LDA + 16 * i];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = bs[4 * ks * LDB + 16 * j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };
    Pair v0 = fetch();
    Pair v1 = fetch();      // past the end: clamped re-loads, staged as zeros
    stage(0, v0, kbeg);
    __syncthreads();
    v0 = fetch();
    const int pairs = ((kend - kbeg + WG_K - 1) / WG_K + 1) / 2;
    int k0 = kbeg;
    for (int p = 0; p < pairs; ++p, k0 += 2 * WG_K) {
        __builtin_amdgcn_sched_barrier(0);   // keep the requests in front of the MFMAs
        multiply(0);
        stage(1, v1, k0 + WG_K);
        __syncthreads();
        v1 = fetch();
        __builtin_amdgcn_sched_barrier(0);
        multiply(1);
        stage(0, v0, k0 + 2 * WG_K);
        __syncthreads();
        v0 = fetch();
    }
    float* pz = jb.part + (long)bz * M * N;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + 32 * wc + 16 * j + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 64 * wr + 16 * i + 4 * (lane >> 4) + r;
                if (m < M && n < N) pz[(long)m * N + n] = acc[i][j][r];
            }
        }
}
inline int wgrad_job_blocks(const WgradJob& j) { return j.gx * j.gy * j.gz; }

// launches what wgrad_gemm(..., jobs) collected
inline void wgrad_jobs_flush(hipStream_t s, WgradJobs& jobs) {
    if (jobs.count == 0) return;
    std::stable_sort(jobs.j, jobs.j + jobs.count, [](const WgradJob& a, const WgradJob& b) { return (long)a.kps * a.tn > (long)b.kps * b.tn; });
    int blocks = 0;
    for (int q = 0; q < jobs.count; ++q) blocks += jobs.j[q].gx * jobs.j[q].gy * jobs.j[q].gz;
    HowlProfScope prof("gemm", s, jobs.flops);
    hipLaunchKernelGGL(wgrad_big_multi_kernel, dim3(blocks), dim3(WG_THREADS), 0, s, jobs);
    jobs.count = 0;
    jobs.flops = 0.0;
}

// fewer rows than this: the 64 x 64 split-K path (HOWL_WGRAD_BIG_MIN_ROWS overrides: the emulator tests run small shapes)
inline int wgrad_big_min_rows() {
    const char* e = getenv("HOWL_WGRAD_BIG_MIN_ROWS");
    return e != nullptr ? atoi(e) : 2048;
}

// dW (N_out, K_in) = dOut^T (N_out x rows) . In (rows x K_in), rows given by row maps; split-K + deterministic sum
// (scratch: splits x N_out x K_in floats, splits = clamp(rows / 512, 1, max_splits))
void wgrad_gemm(hipStream_t s, const float* dout, RowMap dm, int n_out, const float* in, RowMap im, int k_in, int rows,
                float* scratch, float* dw, int max_splits = 64, int rows_per_split = 512, SlabSums* defer = nullptr,
                WgradJobs* jobs = nullptr) {
    int splits = rows / rows_per_split;
    splits = splits < 1 ? 1 : (splits > max_splits ? max_splits : splits);
    if (n_out <= 8 && ((n_out & 3) != 0 || (k_in & 3) != 0)) {   // thin output that the vector GEMM cannot take
        int kp = 64;
        if (k_in <= 32) {
            kp = 1;
            while (kp < k_in) kp <<= 1;
        }
        const int rpc = (rows + splits - 1) / splits;
        const int chunks = (rows + rpc - 1) / rpc;
        const dim3 grid((k_in + 63) / 64, chunks);
#define HOWL_THIN(NO) \
    case NO: hipLaunchKernelGGL(thin_wgrad_kernel<NO>, grid, dim3(256), 0, s, dout, dm, in, im, k_in, kp, rows, rpc, scratch); break;
        switch (n_out) {
            HOWL_THIN(1) HOWL_THIN(2) HOWL_THIN(3) HOWL_THIN(4) HOWL_THIN(5) HOWL_THIN(6) HOWL_THIN(7) HOWL_THIN(8)
        }
#undef HOWL_THIN
        const long n = (long)n_out * k_in;
        if (defer != nullptr) {
            defer->add(scratch, chunks, n, dw);
            return;
        }
        hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, (const float*)scratch, chunks, n, dw);
        return;
    }
    int z;
    auto map4 = [](const RowMap& r) { return (r.s_outer & 3) == 0 && (r.s_inner & 3) == 0; };
    auto al16 = [](const float* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (n_out >= 128 && rows >= wgrad_big_min_rows() && (n_out & 3) == 0 && (k_in & 3) == 0 && k_in >= 16 && map4(dm) && map4(im) && al16(dout) &&
        al16(in) && max_splits >= 16) {
        // wide layer, many rows: 128-row tiles, split count = one block per CU (two when the slabs stay small)
        const int tn = k_in > 64 ? 128 : 64;
        const int tiles = ((n_out + WG_M - 1) / WG_M) * ((k_in + tn - 1) / tn);
        int sp = std::max(1, howl_num_cus() / tiles);
        sp = std::min(sp, max_splits);
        int kps = ((rows + sp - 1) / sp + WG_K - 1) / WG_K * WG_K;
        kps = std::max(kps, 4 * WG_K);
        z = (rows + kps - 1) / kps;
        const dim3 grid((k_in + tn - 1) / tn, (n_out + WG_M - 1) / WG_M, z);
        const bool klin = is_lin(dm) && is_lin(im);
        if (jobs != nullptr && defer != nullptr && jobs->count < MAX_WGRAD_JOBS) {     // launched by wgrad_jobs_flush with its companions
            jobs->j[jobs->count++] = WgradJob{dout, dm, in, im, n_out, k_in, rows, kps, scratch, (int)grid.x, (int)grid.y, (int)grid.z, tn, klin ? 1 : 0,
                                              WgradSecond{nullptr, RowMap{1, 0, 0}, 0, nullptr}};
            jobs->flops += 2.0 * (double)n_out * k_in * rows;
            defer->add(scratch, z, (long)n_out * k_in, dw);
            return;
        }
        HowlProfScope prof("gemm", s, 2.0 * (double)n_out * k_in * rows);
        if (tn == 128) {
            if (klin) hipLaunchKernelGGL((wgrad_big_kernel<128, true>), grid, dim3(WG_THREADS), 0, s, dout, dm, in, im, n_out, k_in, rows, kps, scratch);
            else hipLaunchKernelGGL((wgrad_big_kernel<128, false>), grid, dim3(WG_THREADS), 0, s, dout, dm, in, im, n_out, k_in, rows, kps, scratch);
        } else {
            if (klin) hipLaunchKernelGGL((wgrad_big_kernel<64, true>), grid, dim3(WG_THREADS), 0, s, dout, dm, in, im, n_out, k_in, rows, kps, scratch);
            else hipLaunchKernelGGL((wgrad_big_kernel<64, false>), grid, dim3(WG_THREADS), 0, s, dout, dm, in, im, n_out, k_in, rows, kps, scratch);
        }
    } else {
        // A(m = out col, k = row) = dout[dm(k) + m]  -> unit stride is m
        z = gemm(s, false, dout, lin(1), 0, dm, in, im, 1, n_out, k_in, rows, splits, nullptr, 0, scratch, k_in, (long)n_out * k_in);
    }
    const long n = (long)n_out * k_in;
    if (defer != nullptr) {
        defer->add(scratch, z, n, dw);
        return;
    }
    hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, (const float*)scratch, z, n, dw);
}

// dW (N_out, 128) = dOut^T In and dW2 (N_out, k2) = dOut^T In2 over the same `rows` mapped rows as ONE job of the multi-job launch
// (dout read once for both): wide outputs, many rows, k2 <= 64.  Returns false when the shape does not fit (the caller then runs
// the two products as separate wgrad_gemm jobs).
bool wgrad_dual_gemm(const float* dout, RowMap dm, int n_out, const float* in, RowMap im, const float* in2, RowMap im2, int k2,
                     int rows, float* scratch, float* dw, float* scratch2, float* dw2, int max_splits, SlabSums* defer,
                     WgradJobs* jobs) {
    auto map4 = [](const RowMap& r) { return (r.s_outer & 3) == 0 && (r.s_inner & 3) == 0; };
    auto al16 = [](const float* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (jobs == nullptr || defer == nullptr || jobs->count >= MAX_WGRAD_JOBS || n_out < 128 || (n_out & 3) != 0 ||
        rows < wgrad_big_min_rows() || k2 < 4 || k2 > 64 || (k2 & 3) != 0 || !map4(dm) || !map4(im) || !map4(im2) || !al16(dout) ||
        !al16(in) || !al16(in2) || max_splits < 16 || getenv("HOWL_WGRAD_NO_DUAL") != nullptr)
        return false;
    const int tiles = (n_out + WG_M - 1) / WG_M;
    int sp = std::max(1, howl_num_cus() / tiles);
    sp = std::min(sp, max_splits);
    int kps = ((rows + sp - 1) / sp + WG_K - 1) / WG_K * WG_K;
    kps = std::max(kps, 4 * WG_K);
    const int z = (rows + kps - 1) / kps;
    jobs->j[jobs->count++] = WgradJob{dout, dm, in, im, n_out, 128, rows, kps, scratch, 1, tiles, z, 192, 0, WgradSecond{in2, im2, k2, scratch2}};
    jobs->flops += 2.0 * (double)n_out * (128 + k2) * rows;
    defer->add(scratch, z, (long)n_out * 128, dw);
    defer->add(scratch2, z, (long)n_out * k2, dw2);
    return true;
}

// out0 (and out1) = column sums of x over `rows` mapped rows; scratch holds <= 64 * n floats
void colsum(hipStream_t s, const float* x, RowMap rm, int rows, int n, float* scratch, float* out0, float* out1,
            int max_chunks = 64, int rows_per_chunk = 256, SlabSums* defer = nullptr) {
    int chunks = rows / rows_per_chunk;
    chunks = chunks < 1 ? 1 : (chunks > max_chunks ? max_chunks : chunks);   // scratch holds <= max_chunks slabs of n floats
    const int rpc = (rows + chunks - 1) / chunks;
    hipLaunchKernelGGL(colsum_kernel, dim3((n + 63) / 64, chunks), dim3(256), 0, s, x, rm, rows, n, rpc, scratch);
    if (defer != nullptr) {
        defer->add(scratch, chunks, n, out0, out1);
        return;
    }
    hipLaunchKernelGGL(sum_slabs_kernel, dim3((n + 63) / 64), dim3(256), 0, s, (const float*)scratch, chunks, (long)n, out0);
    if (out1 != nullptr)
        hipLaunchKernelGGL(sum_slabs_kernel, dim3((n + 63) / 64), dim3(256), 0, s, (const float*)scratch, chunks, (long)n, out1);
}


}  // namespace
