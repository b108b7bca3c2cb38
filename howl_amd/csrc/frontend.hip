// Audio frontend kernels for gfx950 (MI355X): fused windowed-FFT + mel + log (K1), deltas (K2),
// ZMUV statistics (K4), SpecAugment masks (K5), mel-filterbank packing / VTLP construction (K3).
//
// Replaces the ATen/torchaudio op chain the reference runs at
//   howl/data/transform/transform.py:249-254,271-280  (MelSpectrogram -> +1e-7 -> log -> ComputeDeltas x2)
//   howl/data/transform/transform.py:373-410,429-449  (VTLP filterbank)
//   howl/data/transform/operator.py:119-146           (ZmuvTransform)
//   howl/data/transform/transform.py:299-339          (SpecAugmentTransform masks)
//
// K1 design (one launch for the whole batch, one persistent workgroup per CU, every WAVEFRONT an independent worker):
//  * frames are flattened over (utterance, t): g = b*T + t; a wave owns "quads" of 4 consecutive frames and runs one real-input
//    FFT-512 per frame as a complex FFT-256 of z[n] = x[2n] + i x[2n+1] on 16 lanes x 16 points, 4 frames side by side;
//  * FFT-256 = 16 x 16: lane (frame i, n2) loads its 16 complex points as sixteen 8-byte loads at immediate offsets (a frame
//    row of 16 lanes reads 128 contiguous bytes; torch.stft's centre / reflect padding only costs index arithmetic on the
//    quads that touch an utterance edge), multiplies the window, runs a 16-point DFT entirely in registers (radix 4 x 4,
//    compile-time twiddles), applies W_256^(n2 k1) and hands the result to lane (k1, frame i) through ONE transpose in the
//    wave's private LDS tile (no workgroup barrier, no wait: a wave's DS operations execute in order); the second in-register
//    DFT-16 leaves Z[j + 16 k2], k2 = 0..15, in lane 4j + i;
//  * real-input recombination without another exchange: lane class j pairs bin k = j + 16 r with 256 - k, whose Z lives in
//    lane class 16 - j; eight ds_bpermute pairs fetch them, and each lane produces |X|^2 for its 16 bins
//    (slot r: bin j + 16 r, slot 8 + r: bin 256 - j - 16 r; bin 128 rides in a 17th slot of class 0);
//  * the power values never leave the registers: with lane = 4j + i they ARE the A operand of v_mfma_f32_4x4x1_16b_f32
//    (block j, row = frame i), 16 independent 4x4 outer products per instruction against filterbank fragments
//    (lane 4j + c = fb[bin(j, slot)][4g + c]) staged once per workgroup in LDS; only the (slot, mel group) pairs the HTK
//    triangles can touch (standard or VTLP-warped: 49 of 170) are multiplied, a flag built with the fragments sends any
//    other matrix through all pairs;
//  * the 16 blocks are summed by a reduce-scatter (v_permlane32_swap, v_permlane16_swap, two DPP levels) that leaves 2-3
//    (frame, mel) sums per lane; log(x + eps), optional ZMUV, stores as (B,T,M) [model layout] or (B,M,T).
// 132 KB of LDS per 12-wave workgroup (9.1 KB transpose tile per wave + 23 KB of tables), 126 VGPRs.
// Algorithmic HBM bytes per utterance: 4*L read + 4*M*T written (76,960 B at L=16000, M=40).
#include <stdlib.h>
#include <type_traits>

#include "howl_common.hip.h"
#include "howl_tables.h"
#include "../../include/howl_hip.h"

namespace {

constexpr int N_FFT = 512;
constexpr int HOP = 200;
constexpr int N_FREQ = 257;
constexpr int K_PAD = 260;          // rows of the row-major packed filterbank (257 padded to a multiple of 4)
constexpr int QUAD = 4;             // frames per wave iteration == rows of a 4x4x1 MFMA block
constexpr int NSLOT = 17;           // power values per lane: 16 bins of its class + bin 128 (class 0 only)
constexpr int NG_MAX = HOWL_FB_COLS / 4;   // mel groups of 4 (columns of a block): 12
constexpr int NG_BANDED = 10;       // the banded fragment table covers 40 mel bins
// packed filterbank buffer: [ (260, 48) row-major | banded fragments [17][64][4] | dense fragments [17][12][64] | 32 ints ]
constexpr int FBQ_OFF = K_PAD * HOWL_FB_COLS;
constexpr int FBQ_FLOATS = NSLOT * 64 * 4;
constexpr int FBD_OFF = FBQ_OFF + FBQ_FLOATS;
constexpr int FBD_FLOATS = NSLOT * NG_MAX * 64;
constexpr int FBF_OFF = FBD_OFF + FBD_FLOATS;     // [0]: 1 when every non-zero weight is covered by the banded table
static_assert(FBF_OFF + 32 == HOWL_FB_PACKED_FLOATS, "include/howl_hip.h and the kernels disagree on the packed filterbank size");
constexpr int C_WIN = 0, C_TW = 16 * HOWL_FE_WIN_PITCH, C_PT = 32 * HOWL_FE_WIN_PITCH;
static_assert(C_PT + 16 * HOWL_FE_PT_PITCH == HOWL_FE_CONST_FLOATS && HOWL_FE_CONST_FLOATS % 4 == 0, "constant table layout");

// Bin held by lane class j (= lane >> 2) in power slot s; -1: the slot is empty for this class.
__host__ __device__ constexpr int bin_of(int s, int j) {
    return s < 8 ? j + 16 * s : (s < 16 ? 256 - j - 16 * (s - 8) : (j == 0 ? 128 : -1));
}
// Inverse: bin k -> slot (class = class_of_bin).
__host__ __device__ constexpr int slot_of_bin(int k) { return k < 128 ? (k >> 4) : (k == 128 ? 16 : 8 + ((256 - k) >> 4)); }
// The mel groups (of 4 bins) a slot can reach: union over the standard HTK filterbank (40 mels, 0-8 kHz, 257 bins) and its
// VTLP warps for alpha in [0.9, 1.1] (transform.py:373-410, the alpha > 1 re-mask quirk included; swept offline).  Component
// q of the slot's 16-byte fragment entry belongs to group slot_group(s, q); -1 = unused.
__host__ __device__ constexpr int slot_group(int s, int q) {
    constexpr signed char t[NSLOT][4] = {{0, 1, 2, 8},   {1, 2, 3, 8},   {3, 4, 8, -1},  {4, 5, 8, -1},  {4, 5, 6, 8},  {5, 6, 7, 8},
                                         {6, 7, 8, -1},  {6, 7, 8, -1},  {7, 9, -1, -1}, {7, 9, -1, -1}, {7, 9, -1, -1}, {7, 8, 9, -1},
                                         {7, 8, 9, -1},  {7, 8, 9, -1},  {7, 8, -1, -1}, {7, 8, -1, -1}, {7, 8, -1, -1}};
    return t[s][q];
}
__host__ __device__ constexpr bool slot_has_group(int s, int g) {
    return slot_group(s, 0) == g || slot_group(s, 1) == g || slot_group(s, 2) == g || slot_group(s, 3) == g;
}

// ---- packed complex arithmetic: a complex number is a register pair (re, im); every helper is ONE v_pk_* instruction whose
// operand modifiers (op_sel: which half feeds which result half; neg_lo / neg_hi) do the swaps and sign flips.  The clean
// cases are plain vector C++ (v_pk_add_f32 / v_pk_mul_f32); the ones with modifiers are spelled out, because the compiler
// otherwise builds the swizzled operand with v_mov / v_pk_mov first (a wave issues one instruction per ~4.4 cycles whatever
// its kind: measured with tools/valu_ubench.hip, so every instruction saved is time saved).
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
#if defined(HIPEMU)
__device__ __forceinline__ v2f cadd_mi(v2f a, v2f b) { return v2f{a.x + b.y, a.y - b.x}; }          // a + (-i) b
__device__ __forceinline__ v2f csub_mi(v2f a, v2f b) { return v2f{a.x - b.y, a.y + b.x}; }          // a - (-i) b
__device__ __forceinline__ v2f cmul(v2f a, v2f w) { return v2f{fmaf(-a.y, w.y, a.x * w.x), fmaf(a.x, w.y, a.y * w.x)}; }
__device__ __forceinline__ v2f cmul_s(v2f a, v2f w) { return cmul(a, w); }
__device__ __forceinline__ v2f pk_scale_s(v2f a, v2f w) { return a * w; }
__device__ __forceinline__ v2f recomb_e(v2f zk, v2f zn) { return v2f{zk.x + zn.x, zk.y - zn.y}; }   // Z[k] + conj Z[256-k]
__device__ __forceinline__ v2f recomb_o(v2f zk, v2f zn) { return v2f{zk.y + zn.y, zn.x - zk.x}; }   // (Z[k] - conj Z[256-k]) / i
#else
__device__ __forceinline__ v2f cadd_mi(v2f a, v2f b) {
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ v2f csub_mi(v2f a, v2f b) {
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// (a.x w.x - a.y w.y, a.y w.x + a.x w.y): the twiddle pair w = (re, im) in VGPRs (cmul) or SGPRs (cmul_s)
__device__ __forceinline__ v2f cmul(v2f a, v2f w) {
    v2f t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(a), "v"(w), "v"(t));
    return d;
}
__device__ __forceinline__ v2f cmul_s(v2f a, v2f w) {
    v2f t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(a), "s"(w), "v"(t));
    return d;
}
__device__ __forceinline__ v2f pk_scale_s(v2f a, v2f w) {
    v2f d;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "s"(w));
    return d;
}
__device__ __forceinline__ v2f recomb_e(v2f zk, v2f zn) {
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(d) : "v"(zk), "v"(zn));
    return d;
}
__device__ __forceinline__ v2f recomb_o(v2f zk, v2f zn) {
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(d) : "v"(zk), "v"(zn));
    return d;
}
#endif

// 4-point forward DFT in place, natural order; ROT2: input 2 still carries a pending factor -i
template <bool ROT2 = false>
__device__ __forceinline__ void dft4(v2f& a0, v2f& a1, v2f& a2, v2f& a3) {
    const v2f s0 = ROT2 ? cadd_mi(a0, a2) : a0 + a2, d0 = ROT2 ? csub_mi(a0, a2) : a0 - a2;
    const v2f s1 = a1 + a3, d1 = a1 - a3;
    a0 = s0 + s1;
    a2 = s0 - s1;
    a1 = cadd_mi(d0, d1);
    a3 = csub_mi(d0, d1);
}

// 16-point forward DFT in registers: v[k] = sum_n v[n] exp(-2 pi i n k / 16), natural order in and out.  Radix 4 x 4 with
// n = 4p + q, k = ka + 4 kb; 32 + 16 + 32 packed instructions.
__device__ __forceinline__ void dft16(v2f (&v)[16]) {
    const float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f, H = 0.70710678118654752440f;
    const v2f W1 = {C1, -S1}, W3 = {S1, -C1}, W9 = {-C1, S1}, HP = {H, H}, HN = {-H, -H};
#pragma unroll
    for (int q = 0; q < 4; ++q) dft4(v[q], v[4 + q], v[8 + q], v[12 + q]);
    // position 4 ka + q now holds B_q[ka]; multiply by W16^(q ka)
    v[5] = cmul_s(v[5], W1);                              // W^1
    v[6] = pk_scale_s(cadd_mi(v[6], v[6]), HP);           // W^2 = H (1 - i): (x + y, y - x) H
    v[7] = cmul_s(v[7], W3);                              // W^3
    v[9] = pk_scale_s(cadd_mi(v[9], v[9]), HP);           // W^2
    //   v[10] * W^4 = -i v[10]: folded into the consuming butterfly (ROT2)
    v[11] = pk_scale_s(csub_mi(v[11], v[11]), HN);        // W^6 = H (-1 - i): (x - y, x + y) (-H)
    v[13] = cmul_s(v[13], W3);                            // W^3
    v[14] = pk_scale_s(csub_mi(v[14], v[14]), HN);        // W^6
    v[15] = cmul_s(v[15], W9);                            // W^9 = -W^1
    dft4(v[0], v[1], v[2], v[3]);
    dft4(v[4], v[5], v[6], v[7]);
    dft4<true>(v[8], v[9], v[10], v[11]);
    dft4(v[12], v[13], v[14], v[15]);
    // position 4 ka + kb holds X[ka + 4 kb]: transpose the 4 x 4 index grid (register renaming)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a + 1; b < 4; ++b) {
            const v2f t = v[4 * a + b];
            v[4 * a + b] = v[4 * b + a];
            v[4 * b + a] = t;
        }
}

// ---- cross-lane primitives of the block reduction (each returns the sum over one bit of the lane index and leaves the
// sum of `lo` in the lanes where that bit is 0, the sum of `hi` where it is 1) ----------------------------------------------
__device__ __forceinline__ float fold_bit5(float lo, float hi) {        // lanes l <-> l ^ 32
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float fold_bit4(float lo, float hi) {        // lanes l <-> l ^ 16
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float fold_bit3(float lo, float hi, bool bit) {   // lanes l <-> l ^ 8 (row_ror:8)
    const float give = bit ? lo : hi, keep = bit ? hi : lo;
    return keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(give), 0x128, 0xf, 0xf, true));
}
__device__ __forceinline__ float fold_bit2(float lo, float hi, bool bit) {   // lanes l <-> l ^ 4 (row_half_mirror, then quad reverse)
    const float give = bit ? lo : hi, keep = bit ? hi : lo;
    const int m = __builtin_amdgcn_update_dpp(0, __float_as_int(give), 0x141, 0xf, 0xf, true);
    return keep + __int_as_float(__builtin_amdgcn_update_dpp(0, m, 0x1b, 0xf, 0xf, true));
}

// Table reads inside the persistent loop are loop-invariant; left alone, the compiler hoists all of them (window, twiddles,
// fragments: 150+ registers) in front of the loop and spills.  An index laundered once per trip keeps them where they are.
// (HOWL_OPAQUE_V: howl_common.hip.h)
#if defined(HIPEMU)
#define HOWL_OPAQUE_S(x) asm volatile("" : "+r"(x))
#define HOWL_OPAQUE_F(x) asm("" : "+x"(x))
#else
#define HOWL_OPAQUE_F(x) asm("" : "+v"(x))
#define HOWL_OPAQUE_S(x) asm volatile("" : "+s"(x))
#endif

#define HOWL_FE_PROBE(wave_, lane_, slot_) ((void)0)

// Waves per CU and the transpose tile of a wave: [frame][n2 row][k1] complex elements with row pitch XR and frame pitch XF.
// The compiler pairs the 8-byte accesses (ds_write2_b64: 8-lane groups writing 16 contiguous bytes each; ds_read2_b64: 16-lane
// groups, 32 banks), so conflict-free means 2 XR = 4 (mod 32) dwords and XF = 4 (mod 16) elements: 18 / 292, which fits twelve
// waves (three per SIMD); sixteen (four per SIMD, 128 VGPRs) only fit the LDS with 17 / 272 and two- to four-way conflicts on
// the 16 transpose instructions of a quad.
template <int NWAVES>
struct FeGeom {
    static constexpr int XR = NWAVES > 12 ? 17 : 18;
    static constexpr int XF = NWAVES > 12 ? 272 : 292;
};
constexpr int FE_WAVES = 12;     // measured at 512 x 1 s: 23.9 us with twelve waves, 24.7 us with sixteen (HOWL_LOGMEL_WAVES=16)

// NGRP = 10: filterbanks of up to 40 mel bins (banded fragments when the flag allows); 12: up to 48, all pairs.
template <int NWAVES, int NGRP>
__global__ __launch_bounds__(NWAVES * 64) void logmel_kernel(const float* __restrict__ pcm, int L, long ld, int T,
                                                             int total_frames, const float* __restrict__ fbp, int M,
                                                             float log_eps, const float* __restrict__ zmuv,
                                                             float* __restrict__ out, int layout, int n_quads, int aligned) {
    constexpr int XR = FeGeom<NWAVES>::XR, XF = FeGeom<NWAVES>::XF;
    __shared__ v2f xch[NWAVES * QUAD * XF];             // FFT transpose tiles, private per wave
    __shared__ v4f c_tab[HOWL_FE_CONST_FLOATS / 4];     // window | W_256 | W_512 rows (read-only after the prologue)
    __shared__ v4f c_frag[NSLOT * 64];                  // banded filterbank fragments

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i1 = lane >> 4, n2 = lane & 15;              // FFT step 1: frame i1 of the quad, column n2
    const int j = lane >> 2, i2 = lane & 3;                // FFT step 2 and everything after: bin class j, frame i2

    // this workgroup's contiguous share of the quads, dealt round-robin to its waves
    const int q_lo = (int)(((long)n_quads * blockIdx.x) / gridDim.x);
    const int q_hi = (int)(((long)n_quads * (blockIdx.x + 1)) / gridDim.x);
    int q = q_lo + wave;
    // (utterance, t) of the quad's first frame: one scalar division here, then stepping
    int b0 = (QUAD * q) / T, t0 = QUAD * q - b0 * T;
    const int step_b = (QUAD * NWAVES) / T, step_t = QUAD * NWAVES - step_b * T;

    // Raw samples of one quad: lane (i1, n2) takes z[16 n1 + n2] = (x[32 n1 + 2 n2], x[.. + 1]) of frame i1, centre framing with
    // reflect padding (torch.stft(center=True)).  Quads whose four frames lie inside one utterance and need no padding are
    // sixteen 8-byte loads at immediate offsets from one base; the others pay per-sample index arithmetic.
    auto fetch = [&](int qq, int bq, int tq, v2f (&x)[16]) {
        const int g0 = QUAD * qq;
        const bool fast = aligned != 0 && tq + 3 < T && HOP * tq >= N_FFT / 2 && HOP * (tq + 3) + N_FFT / 2 <= L && g0 + 3 < total_frames;
        if (fast) {
            // uniform base (SGPRs) + one 32-bit lane offset + immediates
            const float* base = pcm + ((long)bq * ld + (HOP * tq - N_FFT / 2));
            const unsigned off = (unsigned)(HOP * i1 + 2 * n2);
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) x[n1] = *reinterpret_cast<const v2f*>(base + (off + 32u * n1));
        } else {
            int t = tq + i1, b = bq;
            if (t >= T) { t -= T; ++b; }
            if (t >= T) { t -= T; ++b; }
            const bool valid = g0 + i1 < total_frames;
            if (!valid) b = t = 0;
            const unsigned row = (unsigned)b * (unsigned)ld;       // the host checked that the batch spans < 2^31 samples
            const int s0 = HOP * t - N_FFT / 2 + 2 * n2;
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                int sa = s0 + 32 * n1, sb = sa + 1;
                sa = sa < 0 ? -sa : sa;
                sb = sb < 0 ? -sb : sb;
                sa = sa >= L ? 2 * (L - 1) - sa : sa;
                sb = sb >= L ? 2 * (L - 1) - sb : sb;
                const float va = pcm[row + (unsigned)sa], vb = pcm[row + (unsigned)sb];      // unconditional loads, masked afterwards
                x[n1].x = valid ? va : 0.0f;
                x[n1].y = valid ? vb : 0.0f;
            }
        }
    };

    v2f x0[16];
    if (q < q_hi) fetch(q, b0, t0, x0);

    // ---- workgroup prologue: tables into LDS -----------------------------------------------------------
    {
        const v4f* src = reinterpret_cast<const v4f*>(HOWL_FE_CONST);
        for (int i = tid; i < HOWL_FE_CONST_FLOATS / 4; i += NWAVES * 64) c_tab[i] = src[i];
        const v4f* fq = reinterpret_cast<const v4f*>(fbp + FBQ_OFF);
        for (int i = tid; i < NSLOT * 64; i += NWAVES * 64) c_frag[i] = fq[i];
    }
    const bool banded = NGRP == NG_BANDED && reinterpret_cast<const int*>(fbp + FBF_OFF)[0] != 0;   // wave-uniform
    float zm_mean = 0.0f, zm_rstd = 1.0f;
    if (zmuv != nullptr) {
        zm_mean = zmuv[0];
        zm_rstd = 1.0f / zmuv[1];
    }
    __syncthreads();                                       // the only workgroup barrier of the kernel

    // ---- per-lane constants ----------------------------------------------------------------------------
    v2f* const xw = xch + wave * (QUAD * XF) + i1 * XF + n2 * XR;          // step-1 lane writes row n2: + k1
    const v2f* const xr = xch + wave * (QUAD * XF) + i2 * XF + j;          // step-2 lane reads column j: + n2 * XR
    const int wrow0 = (C_WIN + n2 * HOWL_FE_WIN_PITCH) / 4, trow0 = (C_TW + n2 * HOWL_FE_WIN_PITCH) / 4;
    const int prow0 = (C_PT + j * HOWL_FE_PT_PITCH) / 4;
    const int partner = 4 * (4 * ((16 - j) & 15) + i2);    // byte address of the lane holding Z[256 - k] (ds_bpermute)
    const bool class0 = j == 0;
    const bool bit3 = (lane & 8) != 0, bit2 = (lane & 4) != 0;
    // the mel groups go through the contraction in two passes of NH; after the block reduction a lane owns frame r_out of the
    // quad and <= NE groups of the pass
    constexpr int NH = NGRP / 2;                           // groups per pass
    constexpr int ND = (NH + 1) / 2;                       // values per lane after the bit-3 level
    constexpr int NE = (ND + 1) / 2;                       // ... after the bit-2 level
    const int r_out = ((lane >> 5) & 1) * 2 + ((lane >> 4) & 1);
    const int c_out = lane & 3;
    int pslot = 0;
    HOWL_FE_PROBE(wave, lane, pslot++);   // prologue done

    // The loop carries the WINDOWED samples z of the quad it is about to transform: the raw samples of the next quad are
    // requested at the top of the trip (a whole trip ahead) and multiplied by the window at its bottom,
    // so they are defined and consumed inside one trip -- a loop-carried load result costs a second register set, a copy
    // and a full vmcnt(0) wait at the latch.
    v2f z[16];
    auto apply_window = [&](const v2f (&xs)[16]) {
        int wrow = wrow0;
        HOWL_OPAQUE_V(wrow);
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const v4f w = c_tab[wrow + h];
            z[2 * h] = xs[2 * h] * w.xy;
            z[2 * h + 1] = xs[2 * h + 1] * w.zw;
        }
    };
    if (q < q_hi) apply_window(x0);

    for (; q < q_hi; q += NWAVES) {
        const int g0 = QUAD * q;
        HOWL_FE_PROBE(wave, lane, pslot++);   // quad start
        int trow = trow0, prow = prow0, frow = lane;
        HOWL_OPAQUE_V(trow);
        HOWL_OPAQUE_V(prow);
        HOWL_OPAQUE_V(frow);
        // the wave's next quad
        const int qn = q + NWAVES;
        const bool has_next = qn < q_hi;
        int bn = b0 + step_b, tn = t0 + step_t;
        if (tn >= T) { tn -= T; ++bn; }
        v2f xn[16];
        if (has_next) fetch(qn, bn, tn, xn);
        float P[NSLOT];
        // ---- FFT step 1: DFT-16 over n1 of the windowed samples, twiddle, transpose through LDS ------------------------
        dft16(z);
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const v4f w = c_tab[trow + h];
            if (h > 0) z[2 * h] = cmul(z[2 * h], w.xy);
            z[2 * h + 1] = cmul(z[2 * h + 1], w.zw);
        }
        wave_lds_sync();   // the previous trip's column reads are done
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) xw[k1] = z[k1];
        wave_lds_sync();
        // ---- FFT step 2: lane (j, i2) gathers column k1 = j, DFT-16 over n2 -> Z[j + 16 k2] ------------------------------
#pragma unroll
        for (int n = 0; n < 16; ++n) z[n] = xr[n * XR];
        dft16(z);
        HOWL_FE_PROBE(wave, lane, pslot++);   // transformed
        // ---- real-input recombination: X[k] = E + W O, X[256 - k] = conj(E - W O); powers of both ----------------------
        v2f zn[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            // Z[256 - k] sits in register 15 - r of the partner class (register (16 - r) & 15 of class 0 itself); all sixteen
            // exchanges are issued before the first result is used.  (The values pass through an empty asm: otherwise the
            // select of two array elements becomes one element at a selected index, i.e. a 16-way v_cndmask chain per value.)
            float own_r = z[(16 - r) & 15].x, own_i = z[(16 - r) & 15].y;
            HOWL_OPAQUE_F(own_r);
            HOWL_OPAQUE_F(own_i);
            const float pub_r = class0 ? own_r : z[15 - r].x;
            const float pub_i = class0 ? own_i : z[15 - r].y;
            zn[r].x = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(pub_r)));
            zn[r].y = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(pub_i)));
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const v4f w4 = c_tab[prow + (r >> 1)];
            const v2f e = recomb_e(z[r], zn[r]), o = recomb_o(z[r], zn[r]);
            const v2f t = cmul(o, (r & 1) ? w4.zw : w4.xy);
            const v2f xk = e + t, yk = e - t;
            P[r] = xk.x * xk.x + xk.y * xk.y;
            P[8 + r] = yk.x * yk.x + yk.y * yk.y;
        }
        P[16] = 4.0f * (z[8].x * z[8].x + z[8].y * z[8].y);   // bin 128 = Z[128] itself (class 0; the other classes' weight is 0)
        // ---- mel contraction + block sum + epilogue, in two passes over the mel groups (halves the live accumulators: the whole
        // kernel has to fit the register budget of its wave count) -----------------------------------------------------
        // D_j[frame][mel] += P[frame][bin(j, s)] * fb[bin(j, s)][mel] on 16 independent 4x4 blocks j; lane 4j + c then holds
        // D_j[r][4g + c] in acc[g][r] and the 16 blocks are summed by a reduce-scatter over the lane bits of j:
        // v_permlane32_swap (bit 5: frames 0,1 | 2,3), v_permlane16_swap (bit 4: even | odd frame), two DPP levels (bits 3, 2:
        // which groups), leaving <= NE (frame, group) sums per lane.
        const int g_frame = g0 + r_out;
        long o_base, o_ms;
        if (layout == 1) {
            o_base = (long)g_frame * M;
            o_ms = 1;
        } else {
            int t = t0 + r_out, b = b0;
            if (t >= T) { t -= T; ++b; }
            if (t >= T) { t -= T; ++b; }
            o_base = (long)b * M * T + t;
            o_ms = T;
        }
        auto mel_pass = [&](auto g0c) {
            constexpr int G0 = decltype(g0c)::value;
            f32x4 acc[NH];
#pragma unroll
            for (int g = 0; g < NH; ++g) acc[g] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (banded) {
#pragma unroll
                for (int s = 0; s < NSLOT; ++s) {
                    constexpr int lo = G0, hi = G0 + NH;
                    const bool any = (slot_group(s, 0) >= lo && slot_group(s, 0) < hi) || (slot_group(s, 1) >= lo && slot_group(s, 1) < hi) ||
                                     (slot_group(s, 2) >= lo && slot_group(s, 2) < hi) || (slot_group(s, 3) >= lo && slot_group(s, 3) < hi);
                    if (!any) continue;                            // compile-time after unrolling
                    const v4f f = c_frag[s * 64 + frow];
                    const float fv[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int g = slot_group(s, c);
                        if (g >= lo && g < hi) acc[g - lo] = __builtin_amdgcn_mfma_f32_4x4x1f32(P[s], fv[c], acc[g - lo], 0, 0, 0);
                    }
                }
            } else {
                // the rare path (a matrix the banded table does not cover, or more than 40 mel bins): every (slot, group) pair,
                // fragments from global memory at a uniform base + lane
                const unsigned ulane = (unsigned)frow;
                const float* fdense = fbp + FBD_OFF;
                HOWL_OPAQUE_S(fdense);
#pragma unroll
                for (int s = 0; s < NSLOT; ++s) {
                    const float* fs = fdense + s * (NG_MAX * 64);
#pragma unroll
                    for (int g = 0; g < NH; ++g)
                        acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(P[s], fs[(G0 + g) * 64 + ulane], acc[g], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);   // keep the fragment loads from piling up in registers
                }
            }
            float v2[2 * NH];
#pragma unroll
            for (int e = 0; e < 2 * NH; ++e) v2[e] = fold_bit5(acc[e % NH][e / NH], acc[e % NH][e / NH + 2]);
            float v1[NH];
#pragma unroll
            for (int g = 0; g < NH; ++g) v1[g] = fold_bit4(v2[g], v2[NH + g]);
            float vc[ND];
#pragma unroll
            for (int g = 0; g < ND; ++g) vc[g] = fold_bit3(v1[g], ND + g < NH ? v1[ND + g] : 0.0f, bit3);
            float vd[NE];
#pragma unroll
            for (int g = 0; g < NE; ++g) vd[g] = fold_bit2(vc[g], NE + g < ND ? vc[NE + g] : 0.0f, bit2);
            // log(x + eps), ZMUV, store: this lane's groups of the pass
#pragma unroll
            for (int h = 0; h < NE; ++h) {
                const int w = (bit2 ? NE : 0) + h, u = (bit3 ? ND : 0) + w;      // position at the two DPP levels
                const int m = 4 * (G0 + u) + c_out;
                if (w < ND && u < NH && m < M && g_frame < total_frames) {
                    float y = __builtin_amdgcn_logf(vd[h] + log_eps) * 0.69314718055994530942f;
                    y = (y - zm_mean) * zm_rstd;
                    out[o_base + (long)m * o_ms] = y;
                }
            }
        };
        mel_pass(std::integral_constant<int, 0>{});
        HOWL_FE_PROBE(wave, lane, pslot++);   // contracted (first half)
        mel_pass(std::integral_constant<int, NH>{});
        HOWL_FE_PROBE(wave, lane, pslot++);   // stored
        if (has_next) apply_window(xn);
        b0 = bn;
        t0 = tn;
    }
}

// fb (257, M) row-major -> fbp (260, NCOL) zero padded
__global__ void fb_pack_kernel(const float* __restrict__ fb, int M, float* __restrict__ fbp, int ncol) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= K_PAD * ncol) return;
    const int k = idx / ncol, m = idx - k * ncol;
    fbp[idx] = (k < N_FREQ && m < M) ? fb[k * M + m] : 0.0f;
}

// Second half of a filterbank build (same stream, after the row-major part is written): the fragment-ordered copies read by
// the 4x4x1 MFMAs of logmel_kernel -- lane 4j + c of fragment (slot s, group g) = fb[bin_of(s, j)][4 g + c] -- once as the
// banded LDS image ([s][lane][4]: the four groups slot_group(s, .) of a slot side by side) and once for every (s, g) pair,
// plus the flag that tells the kernel whether the banded image covers every non-zero weight.  One workgroup.
__global__ __launch_bounds__(1024) void fb_fragments_kernel(float* __restrict__ fbp) {
    __shared__ int uncovered;
    const float* rm = fbp;                 // (K_PAD, HOWL_FB_COLS) row-major
    if (threadIdx.x == 0) uncovered = 0;
    __syncthreads();
    float* fq = fbp + FBQ_OFF;
    for (int idx = threadIdx.x; idx < FBQ_FLOATS; idx += blockDim.x) {
        const int c = idx & 3, lane = (idx >> 2) & 63, s = idx >> 8;
        const int g = slot_group(s, c), bin = bin_of(s, lane >> 2);
        fq[idx] = (g >= 0 && bin >= 0) ? rm[bin * HOWL_FB_COLS + 4 * g + (lane & 3)] : 0.0f;
    }
    float* fd = fbp + FBD_OFF;
    for (int idx = threadIdx.x; idx < FBD_FLOATS; idx += blockDim.x) {
        const int lane = idx & 63, pair = idx >> 6;
        const int s = pair / NG_MAX, g = pair - s * NG_MAX, bin = bin_of(s, lane >> 2);
        fd[idx] = bin >= 0 ? rm[bin * HOWL_FB_COLS + 4 * g + (lane & 3)] : 0.0f;
    }
    for (int idx = threadIdx.x; idx < N_FREQ * HOWL_FB_COLS; idx += blockDim.x) {
        const int k = idx / HOWL_FB_COLS, m = idx - k * HOWL_FB_COLS;
        if (rm[idx] != 0.0f && !slot_has_group(slot_of_bin(k), m >> 2)) uncovered = 1;   // benign race: every writer stores 1
    }
    __syncthreads();
    if (threadIdx.x == 0) reinterpret_cast<int*>(fbp + FBF_OFF)[0] = uncovered ? 0 : 1;
}

// triangles from M+2 corner frequencies (already VTLP-warped on the host: 42 scalars), exactly the
// slope arithmetic of transform.py:402-409; all_freqs = linspace(0, sr/2, 257) = k * (sr/2) / 256.
__device__ __forceinline__ float fb_triangle(const HowlMelPoints& pts, int M, float nyquist, int k, int m) {
    if (k < 0 || k >= N_FREQ || m >= M) return 0.0f;
    const float f = (k == N_FREQ - 1) ? nyquist : (float)k * (nyquist / (float)(N_FREQ - 1));
    const float down = (-1.0f * (pts.f[m] - f)) / (pts.f[m + 1] - pts.f[m]);
    const float up = (pts.f[m + 2] - f) / (pts.f[m + 2] - pts.f[m + 1]);
    return fmaxf(0.0f, fminf(down, up));
}

// The whole packed filterbank of howl_fb_from_points in ONE launch (round 4; it was the row-major matrix, then a one-workgroup
// kernel gathering the fragment images from it: 5 + 11 us per VTLP step, i.e. on 75 % of the training steps): every element
// of the three images is computed from the corner points where it is stored, the last block takes the coverage flag.
__global__ __launch_bounds__(256) void fb_from_points_kernel(HowlMelPoints pts, int M, float nyquist, float* __restrict__ fbp) {
    constexpr int N_RM = K_PAD * HOWL_FB_COLS;
    if (blockIdx.x == gridDim.x - 1) {      // does the banded image cover every non-zero weight?
        __shared__ int uncovered;
        if (threadIdx.x == 0) uncovered = 0;
        __syncthreads();
        for (int idx = threadIdx.x; idx < N_FREQ * HOWL_FB_COLS; idx += blockDim.x) {
            const int k = idx / HOWL_FB_COLS, m = idx - k * HOWL_FB_COLS;
            if (fb_triangle(pts, M, nyquist, k, m) != 0.0f && !slot_has_group(slot_of_bin(k), m >> 2)) uncovered = 1;   // benign race
        }
        __syncthreads();
        if (threadIdx.x == 0) reinterpret_cast<int*>(fbp + FBF_OFF)[0] = uncovered ? 0 : 1;
        return;
    }
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < N_RM) {                       // (K_PAD, HOWL_FB_COLS) row-major
        fbp[idx] = fb_triangle(pts, M, nyquist, idx / HOWL_FB_COLS, idx % HOWL_FB_COLS);
        return;
    }
    idx -= N_RM;
    if (idx < FBQ_FLOATS) {                 // banded LDS image [s][lane][4]
        const int c = idx & 3, lane = (idx >> 2) & 63, sl = idx >> 8;
        const int g = slot_group(sl, c), bin = bin_of(sl, lane >> 2);
        fbp[FBQ_OFF + idx] = (g >= 0 && bin >= 0) ? fb_triangle(pts, M, nyquist, bin, 4 * g + (lane & 3)) : 0.0f;
        return;
    }
    idx -= FBQ_FLOATS;
    if (idx < FBD_FLOATS) {                 // every (slot, group) fragment
        const int lane = idx & 63, pair = idx >> 6;
        const int sl = pair / NG_MAX, g = pair - sl * NG_MAX, bin = bin_of(sl, lane >> 2);
        fbp[FBD_OFF + idx] = bin >= 0 ? fb_triangle(pts, M, nyquist, bin, 4 * g + (lane & 3)) : 0.0f;
    }
}

// K2: (B,M,T) raw log-mels -> (B,3,M,T) [log-mel, delta, delta-delta], each optionally ZMUV-normalised.
// ComputeDeltas(win_length=5, mode="replicate") twice; the second pass pads the *delta* row by replication.
__global__ void deltas_kernel(const float* __restrict__ x, long rows, int M, int T, const float* __restrict__ zmuv,
                              float* __restrict__ out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * T) return;
    const long row = idx / T;
    const int t = (int)(idx - row * T);
    const float* xr = x + row * T;
    auto clampt = [T](int u) { return u < 0 ? 0 : (u >= T ? T - 1 : u); };
    auto delta_at = [&](int u) {
        const float a = xr[clampt(u - 2)], b = xr[clampt(u - 1)], c = xr[clampt(u + 1)], d = xr[clampt(u + 2)];
        return (-2.0f * a + -1.0f * b + c + 2.0f * d) / 10.0f;
    };
    const float l = xr[t];
    const float d0 = delta_at(t);
    const float dm2 = delta_at(clampt(t - 2)), dm1 = delta_at(clampt(t - 1));
    const float dp1 = delta_at(clampt(t + 1)), dp2 = delta_at(clampt(t + 2));
    const float dd = (-2.0f * dm2 + -1.0f * dm1 + dp1 + 2.0f * dp2) / 10.0f;
    float mean = 0.0f, sd = 1.0f;
    if (zmuv != nullptr) {
        mean = zmuv[0];
        sd = zmuv[1];
    }
    const long b = row / M;
    const int m = (int)(row - b * M);
    const long plane = (long)M * T;
    float* o = out + b * 3 * plane + (long)m * T + t;
    o[0] = (l - mean) / sd;
    o[plane] = (d0 - mean) / sd;
    o[2 * plane] = (dd - mean) / sd;
}

// K4: sum and sum of squares in fp64, one atomic pair per workgroup
__global__ __launch_bounds__(256) void sum_sumsq_kernel(const float* __restrict__ x, size_t n, double* __restrict__ out2) {
    __shared__ double red[2][4];
    double s = 0.0, q = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double v = (double)x[i];
        s += v;
        q += v * v;
    }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][wave] = s;
        red[1][wave] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&out2[0], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(&out2[1], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

// the masked variant (operator.py:128-130): sums of x*m and (x*m)^2, and the mask's own sum as the element count
__global__ __launch_bounds__(256) void sum_sumsq_masked_kernel(const float* __restrict__ x, const float* __restrict__ m, size_t n,
                                                               double* __restrict__ out3) {
    __shared__ double red[3][4];
    double s = 0.0, q = 0.0, c = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float mv = m[i];
        const double v = (double)(x[i] * mv);      // the reference squares the fp32 product data * mask
        s += v;
        q += v * v;
        c += (double)mv;
    }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    c = wave_sum_d(c);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][wave] = s;
        red[1][wave] = q;
        red[2][wave] = c;
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicAdd(&out3[threadIdx.x], red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// running update of operator.py:133-135 on the device buffers (no host sync); count < 0: taken from sums[2] (mask sum)
__global__ void zmuv_update_kernel(const double* __restrict__ sums, double count, float* total, float* mean, float* mean2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (count < 0.0) count = sums[2] * -count;      // masked: -count = unexpanded / expanded mask size
    const double tot = (double)total[0];
    mean[0] = (float)((sums[0] + (double)mean[0] * tot) / (tot + count));
    mean2[0] = (float)((sums[1] + (double)mean2[0] * tot) / (tot + count));
    total[0] = (float)(tot + count);
}

// (mean, mean2) -> (mean, std) pair consumed by the fused epilogues
__global__ void zmuv_pair_kernel(const float* mean, const float* mean2, float* pair) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float m = mean[0];
    pair[0] = m;
    pair[1] = sqrtf(mean2[0] - m * m);
}

// operator.py:145-146: (x - mean) / std, elementwise
__global__ void zmuv_apply_kernel(const float* __restrict__ x, size_t n, const float* __restrict__ pair,
                                  float* __restrict__ out) {
    const float mean = pair[0], sd = pair[1];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (x[i] - mean) / sd;
}

// K5: zero x[b, :, f0:f0+f, :] and x[b, :, :, t0:t0+t] for every sample (negative width = no mask)
__global__ void specaug_kernel(float* __restrict__ x, int B, int C, int M, int T, long sb, long sc, long sm, long st,
                               const int* __restrict__ f0, const int* __restrict__ f, const int* __restrict__ t0,
                               const int* __restrict__ t) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per = (long)C * M * T;
    if (idx >= (long)B * per) return;
    const int b = (int)(idx / per);
    const long r = idx - (long)b * per;
    const int c = (int)(r / ((long)M * T));
    const int m = (int)((r / T) % M);
    const int tt = (int)(r % T);
    const bool fm = f[b] > 0 && m >= f0[b] && m < f0[b] + f[b];
    const bool tm = t[b] > 0 && tt >= t0[b] && tt < t0[b] + t[b];
    if (fm || tm) x[b * sb + c * sc + m * sm + tt * st] = 0.0f;
}

// Collate + waveform augmentation on the device (operator.py:73-86, transform.py:120-196): gather clip idx[b] from the
// bank, drop `shift[b]` samples from the head (from_head) or the tail (TimeshiftTransform), add white noise
// N(0, sigma[b]) and salt-and-pepper Bern(p/2) - Bern(p/2) (NoiseTransform; each clamped to [-1,1] like the reference),
// zero-pad right to Lout (batchify).  Randomness is a counter-based hash of (seed, b, n): reproducible, order-free.
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ float u01(unsigned long long h) { return ((unsigned)(h >> 40) + 0.5f) * (1.0f / 16777216.0f); }

__global__ __launch_bounds__(256) void collate_augment_kernel(const float* __restrict__ bank, long bank_ld,
                                                              const int* __restrict__ idx, const int* __restrict__ src_len,
                                                              const int* __restrict__ shift, const int* __restrict__ from_head,
                                                              const float* __restrict__ sigma, const float* __restrict__ sp_prob,
                                                              unsigned long long seed, const float* __restrict__ bg,
                                                              long bg_ld, const int* __restrict__ bg_idx,
                                                              const int* __restrict__ bg_off, const float* __restrict__ alpha,
                                                              const int* __restrict__ dst_off, float* __restrict__ out,
                                                              int Lout) {
    const int b = blockIdx.y;
    const int len = src_len[b] - shift[b];
    const int off = from_head[b] ? shift[b] : 0;
    const int d0 = dst_off != nullptr ? dst_off[b] : 0;      // zeros in front of the samples (tensorize rand_append)
    const float* src = bank + (long)idx[b] * bank_ld + off - d0;
    const float sg = sigma[b], pp = sp_prob[b];
    // DatasetMixer runs before the time shift (train.py:218): the background window follows the crop
    const float al = (bg != nullptr) ? alpha[b] : 0.0f;
    const float* bsrc = (al != 0.0f) ? bg + (long)bg_idx[b] * bg_ld + bg_off[b] + off - d0 : nullptr;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < Lout; n += gridDim.x * blockDim.x) {
        float v = 0.0f;
        if (n >= d0 && n < d0 + len) {
            v = src[n];
            if (bsrc != nullptr) v = v * (1.0f - al) + bsrc[n] * al;
            const unsigned long long key = mix64(seed ^ ((unsigned long long)b << 32) ^ (unsigned long long)n);
            if (sg > 0.0f) {
                const float u1 = u01(key), u2 = u01(mix64(key));
                const float z = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);   // Box-Muller
                v = fminf(fmaxf(v + fminf(fmaxf(z * sg, -1.0f), 1.0f), -1.0f), 1.0f);
            }
            if (pp > 0.0f) {
                const unsigned long long k2 = mix64(key ^ 0xD1B54A32D192ED03ull);
                const float salt = u01(k2) < 0.5f * pp ? 1.0f : 0.0f;
                const float pepper = u01(mix64(k2)) < 0.5f * pp ? 1.0f : 0.0f;
                v = fminf(fmaxf(v + (salt - pepper), -1.0f), 1.0f);
            }
        }
        out[(long)b * Lout + n] = v;
    }
}

// Dropout keep-mask (nn.Dropout(0.2) in front of MobileNet's classifier, cnn.py:22 via torchvision): mask[i] = 1 with
// probability 1 - p from the same counter-based generator, one launch instead of torch's rand / compare / cast chain.
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ mask, size_t n, float p, unsigned long long seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        mask[i] = u01(mix64(seed ^ (0xA0761D6478BD642Full * (i + 1)))) >= p ? 1.0f : 0.0f;
}

// Frame-window gather (batchifier.py:56-118 + operator.py:89-109): row b of the (B, Lout) batch is zeros except for
// out[b, dst_off[b] + n] = bank[idx[b]][start[b] + n], n < len[b] -- the window cut of WakeWordFrameBatchifier followed by
// tensorize_audio_data's zero padding on either side (rand_append).  float4 stores; the source offset is arbitrary so
// the loads stay scalar (L2-served: a window is re-read by no one).
__global__ __launch_bounds__(256) void gather_windows_kernel(const float* __restrict__ bank, long bank_ld,
                                                             const int* __restrict__ idx, const int* __restrict__ start,
                                                             const int* __restrict__ len, const int* __restrict__ dst_off,
                                                             float* __restrict__ out, int Lout) {
    const int b = blockIdx.y;
    const int n0 = dst_off[b], n1 = n0 + len[b];
    const float* src = bank + (long)idx[b] * bank_ld + start[b] - n0;
    float* dst = out + (long)b * Lout;
    const int L4 = Lout & ~3;
    for (int n = (blockIdx.x * blockDim.x + threadIdx.x) * 4; n < L4; n += gridDim.x * blockDim.x * 4) {
        float4 v;
        v.x = (n >= n0 && n < n1) ? src[n] : 0.0f;
        v.y = (n + 1 >= n0 && n + 1 < n1) ? src[n + 1] : 0.0f;
        v.z = (n + 2 >= n0 && n + 2 < n1) ? src[n + 2] : 0.0f;
        v.w = (n + 3 >= n0 && n + 3 < n1) ? src[n + 3] : 0.0f;
        if ((((long)b * Lout) & 3) == 0) {
            *reinterpret_cast<float4*>(dst + n) = v;
        } else {
            dst[n] = v.x; dst[n + 1] = v.y; dst[n + 2] = v.z; dst[n + 3] = v.w;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < Lout - L4) {
        const int n = L4 + threadIdx.x;
        dst[n] = (n >= n0 && n < n1) ? src[n] : 0.0f;
    }
}

}  // namespace


extern "C" {

int howl_dropout_mask(float* mask, size_t n, float p, unsigned long long seed, hipStream_t stream) {
    HOWL_REQUIRE(mask, "howl_dropout_mask: null pointer");
    HOWL_REQUIRE(p >= 0.0f && p < 1.0f, "howl_dropout_mask: p=%g outside [0, 1)", p);
    if (n == 0) return HOWL_OK;
    size_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, mask, n, p, seed);
    HOWL_CHECK_LAUNCH("howl_dropout_mask");
    return HOWL_OK;
}

int howl_gather_windows(const float* bank, long bank_ld, const int* idx, const int* start, const int* len,
                        const int* dst_off, int B, int Lout, float* out, hipStream_t stream) {
    HOWL_REQUIRE(bank && idx && start && len && dst_off && out, "howl_gather_windows: null pointer");
    HOWL_REQUIRE(B >= 1 && Lout >= 1, "howl_gather_windows: bad shape");
    HOWL_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "howl_gather_windows: out must be 16-byte aligned");
    int gx = (Lout / 4 + 255) / 256;
    if (gx < 1) gx = 1;
    if (gx > 16) gx = 16;
    hipLaunchKernelGGL(gather_windows_kernel, dim3(gx, B), dim3(256), 0, stream, bank, bank_ld, idx, start, len, dst_off, out,
                       Lout);
    HOWL_CHECK_LAUNCH("howl_gather_windows");
    return HOWL_OK;
}

int howl_collate_augment_window(const float* bank, long bank_ld, const int* idx, const int* src_len, const int* shift,
                                const int* from_head, const float* sigma, const float* sp_prob, unsigned long long seed,
                                const float* bg, long bg_ld, const int* bg_idx, const int* bg_off, const float* alpha,
                                const int* dst_off, int B, int Lout, float* out, hipStream_t stream) {
    HOWL_REQUIRE(bank && idx && src_len && shift && from_head && sigma && sp_prob && out, "howl_collate_augment: null pointer");
    HOWL_REQUIRE(bg == nullptr || (bg_idx && bg_off && alpha), "howl_collate_augment: background given without its parameters");
    HOWL_REQUIRE(B >= 1 && Lout >= 1, "howl_collate_augment: bad shape");
    int gx = (Lout + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(collate_augment_kernel, dim3(gx, B), dim3(256), 0, stream, bank, bank_ld, idx, src_len, shift, from_head,
                       sigma, sp_prob, seed, bg, bg_ld, bg_idx, bg_off, alpha, dst_off, out, Lout);
    HOWL_CHECK_LAUNCH("howl_collate_augment");
    return HOWL_OK;
}

int howl_collate_augment_mix(const float* bank, long bank_ld, const int* idx, const int* src_len, const int* shift,
                             const int* from_head, const float* sigma, const float* sp_prob, unsigned long long seed,
                             const float* bg, long bg_ld, const int* bg_idx, const int* bg_off, const float* alpha, int B,
                             int Lout, float* out, hipStream_t stream) {
    return howl_collate_augment_window(bank, bank_ld, idx, src_len, shift, from_head, sigma, sp_prob, seed, bg, bg_ld, bg_idx,
                                       bg_off, alpha, nullptr, B, Lout, out, stream);
}

int howl_collate_augment(const float* bank, long bank_ld, const int* idx, const int* src_len, const int* shift,
                         const int* from_head, const float* sigma, const float* sp_prob, unsigned long long seed, int B,
                         int Lout, float* out, hipStream_t stream) {
    return howl_collate_augment_mix(bank, bank_ld, idx, src_len, shift, from_head, sigma, sp_prob, seed, nullptr, 0, nullptr,
                                    nullptr, nullptr, B, Lout, out, stream);
}

int howl_fb_pack(const float* fb, int M, float* fbp, hipStream_t stream) {
    HOWL_REQUIRE(fb && fbp, "howl_fb_pack: null pointer");
    HOWL_REQUIRE(M >= 1 && M <= HOWL_MAX_MELS, "howl_fb_pack: M=%d unsupported (1..%d)", M, HOWL_MAX_MELS);
    const int ncol = HOWL_FB_COLS;
    hipLaunchKernelGGL(fb_pack_kernel, dim3((K_PAD * ncol + 255) / 256), dim3(256), 0, stream, fb, M, fbp, ncol);
    hipLaunchKernelGGL(fb_fragments_kernel, dim3(1), dim3(1024), 0, stream, fbp);
    HOWL_CHECK_LAUNCH("howl_fb_pack");
    return HOWL_OK;
}

int howl_fb_from_points(const HowlMelPoints* pts, int M, float nyquist, float* fbp, hipStream_t stream) {
    HOWL_REQUIRE(pts && fbp, "howl_fb_from_points: null pointer");
    HOWL_REQUIRE(M >= 1 && M <= HOWL_MAX_MELS, "howl_fb_from_points: M=%d unsupported (1..%d)", M, HOWL_MAX_MELS);
    const int ncol = HOWL_FB_COLS;
    (void)ncol;
    const int work = K_PAD * HOWL_FB_COLS + FBQ_FLOATS + FBD_FLOATS;
    hipLaunchKernelGGL(fb_from_points_kernel, dim3((work + 255) / 256 + 1), dim3(256), 0, stream, *pts, M, nyquist, fbp);
    HOWL_CHECK_LAUNCH("howl_fb_from_points");
    return HOWL_OK;
}

int howl_logmel_fwd(const float* pcm, int B, int L, long ld, const float* fbp, int M, float log_eps, const float* zmuv,
                    float* out, int layout, hipStream_t stream) {
    HOWL_REQUIRE(pcm && fbp && out, "howl_logmel_fwd: null pointer");
    HOWL_REQUIRE(B >= 1, "howl_logmel_fwd: empty batch");
    HOWL_REQUIRE(L > N_FFT / 2, "howl_logmel_fwd: L=%d too short for reflect padding (needs > 256, as torch.stft)", L);
    HOWL_REQUIRE(ld >= 0, "howl_logmel_fwd: negative row stride %ld", ld);  // rows may overlap (strided windows of one clip)
    HOWL_REQUIRE(M >= 1 && M <= HOWL_MAX_MELS, "howl_logmel_fwd: M=%d unsupported (1..%d)", M, HOWL_MAX_MELS);
    HOWL_REQUIRE(layout == 0 || layout == 1, "howl_logmel_fwd: layout must be 0 (B,M,T) or 1 (B,T,M)");
    const int T = 1 + L / HOP;
    HOWL_REQUIRE((long)(B - 1) * ld + L < (1L << 31), "howl_logmel_fwd: the batch spans %ld samples (32-bit sample offsets)", (long)(B - 1) * ld + L);
    HOWL_REQUIRE((long)B * T < (1L << 31) - 4096L, "howl_logmel_fwd: B*T = %ld frames exceeds the 32-bit frame index", (long)B * T);
    const int total = B * T;
    const int n_quads = (total + QUAD - 1) / QUAD;
    // one persistent workgroup per CU (its LDS holds the tables once); each takes a contiguous share of the quads
    int grid = howl_num_cus();
    if (grid > n_quads) grid = n_quads;
    // 8-byte sample loads need even row strides and an 8-byte aligned base; anything else takes the per-sample path
    const int aligned = ((ld & 1) == 0 && (reinterpret_cast<uintptr_t>(pcm) & 7) == 0) ? 1 : 0;
    int waves = FE_WAVES;
    if (const char* e = getenv("HOWL_LOGMEL_WAVES")) waves = atoi(e) == 12 ? 12 : 16;   // occupancy experiments
    {
        HowlProfScope prof("logmel", stream);
        if (M > 4 * NG_BANDED)
            hipLaunchKernelGGL((logmel_kernel<12, NG_MAX>), dim3((unsigned)grid), dim3(12 * 64), 0, stream, pcm, L, ld, T, total, fbp, M,
                               log_eps, zmuv, out, layout, n_quads, aligned);
        else if (waves == 16)
            hipLaunchKernelGGL((logmel_kernel<16, NG_BANDED>), dim3((unsigned)grid), dim3(16 * 64), 0, stream, pcm, L, ld, T, total, fbp,
                               M, log_eps, zmuv, out, layout, n_quads, aligned);
        else
            hipLaunchKernelGGL((logmel_kernel<12, NG_BANDED>), dim3((unsigned)grid), dim3(12 * 64), 0, stream, pcm, L, ld, T, total, fbp,
                               M, log_eps, zmuv, out, layout, n_quads, aligned);
    }
    HOWL_CHECK_LAUNCH("howl_logmel_fwd");
    return HOWL_OK;
}

int howl_deltas_fwd(const float* logmel, int B, int M, int T, const float* zmuv, float* out3, hipStream_t stream) {
    HOWL_REQUIRE(logmel && out3, "howl_deltas_fwd: null pointer");
    HOWL_REQUIRE(B >= 1 && M >= 1 && T >= 1, "howl_deltas_fwd: bad shape");
    const long rows = (long)B * M;
    const long n = rows * T;
    hipLaunchKernelGGL(deltas_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, logmel, rows, M, T, zmuv,
                       out3);
    HOWL_CHECK_LAUNCH("howl_deltas_fwd");
    return HOWL_OK;
}

int howl_zmuv_update(const float* x, size_t n, float* total, float* mean, float* mean2, double* scratch2,
                     hipStream_t stream) {
    HOWL_REQUIRE(x && total && mean && mean2 && scratch2, "howl_zmuv_update: null pointer");
    HOWL_REQUIRE(n >= 1, "howl_zmuv_update: empty input");
    hipMemsetAsync(scratch2, 0, 2 * sizeof(double), stream);
    size_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(sum_sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, n, scratch2);
    hipLaunchKernelGGL(zmuv_update_kernel, dim3(1), dim3(64), 0, stream, scratch2, (double)n, total, mean, mean2);
    HOWL_CHECK_LAUNCH("howl_zmuv_update");
    return HOWL_OK;
}

int howl_zmuv_update_masked(const float* x, const float* mask, size_t n, double count_scale, float* total, float* mean,
                            float* mean2, double* scratch3, hipStream_t stream) {
    HOWL_REQUIRE(count_scale > 0.0 && count_scale <= 1.0, "howl_zmuv_update_masked: count_scale %g outside (0, 1]", count_scale);
    HOWL_REQUIRE(x && mask && total && mean && mean2 && scratch3, "howl_zmuv_update_masked: null pointer");
    HOWL_REQUIRE(n >= 1, "howl_zmuv_update_masked: empty input");
    hipMemsetAsync(scratch3, 0, 3 * sizeof(double), stream);
    size_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(sum_sumsq_masked_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, mask, n, scratch3);
    hipLaunchKernelGGL(zmuv_update_kernel, dim3(1), dim3(64), 0, stream, scratch3, -count_scale, total, mean, mean2);
    HOWL_CHECK_LAUNCH("howl_zmuv_update_masked");
    return HOWL_OK;
}

int howl_zmuv_pair(const float* mean, const float* mean2, float* pair, hipStream_t stream) {
    HOWL_REQUIRE(mean && mean2 && pair, "howl_zmuv_pair: null pointer");
    hipLaunchKernelGGL(zmuv_pair_kernel, dim3(1), dim3(64), 0, stream, mean, mean2, pair);
    HOWL_CHECK_LAUNCH("howl_zmuv_pair");
    return HOWL_OK;
}

int howl_zmuv_apply(const float* x, size_t n, const float* pair, float* out, hipStream_t stream) {
    HOWL_REQUIRE(x && pair && out, "howl_zmuv_apply: null pointer");
    if (n == 0) return HOWL_OK;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(zmuv_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, n, pair, out);
    HOWL_CHECK_LAUNCH("howl_zmuv_apply");
    return HOWL_OK;
}

int howl_specaug_mask(float* x, int B, int C, int M, int T, long sb, long sc, long sm, long st, const int* f0, const int* f,
                      const int* t0, const int* t, hipStream_t stream) {
    HOWL_REQUIRE(x && f0 && f && t0 && t, "howl_specaug_mask: null pointer");
    HOWL_REQUIRE(B >= 1 && C >= 1 && M >= 1 && T >= 1, "howl_specaug_mask: bad shape");
    const long n = (long)B * C * M * T;
    hipLaunchKernelGGL(specaug_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, B, C, M, T, sb, sc, sm, st,
                       f0, f, t0, t);
    HOWL_CHECK_LAUNCH("howl_specaug_mask");
    return HOWL_OK;
}

}  // extern "C"
