// The fused log-mel frontend's per-workgroup body and everything it needs (constants, packed complex arithmetic, the in-register
// DFT-16, the cross-lane folds), shared by frontend.hip (logmel_kernel) and -- round 5 -- by kernels that run it as rider blocks.
// Moved out of frontend.hip verbatim; see that file's header for what it replaces in the reference.
#pragma once
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
constexpr int FBF_OFF = FBD_OFF + FBD_FLOATS;     // [0]: 1 when every non-zero weight is covered by the banded table (slot_group layout of
                                                  // FBQ); [1]: the same for a bank of the 80-bin filterbank (wide_mask, pair-major FBQ)
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
// The stock NUM_MELS = 80 (settings.py:32) as two banks of 40 columns, [0, 40) and [40, 80): which of a bank's ten mel groups a
// slot's 16 bins can reach -- the same sweep (standard filterbank + VTLP warps over alpha in [0.9, 1.1] in steps of 2.5e-5, the
// alpha > 1 re-mask quirk included: it is what puts groups 5 / 6 of the upper bank into every slot), as one bit mask per slot.
// 16 + 62 (slot, group) pairs against 2 x 170 for all pairs.  Up to six groups per slot, so the fragments of these tables are
// stored one (slot, group) pair after the other ([pair][64 lanes], wide_pair) instead of four groups side by side.
constexpr int WIDE_MELS = 80, WIDE_BANK = 40;
__host__ __device__ constexpr unsigned wide_mask(int bank, int s) {
    constexpr unsigned short t[2][NSLOT] = {
        {0x01f, 0x0f8, 0x3c0, 0x300, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
        {0x060, 0x060, 0x060, 0x063, 0x067, 0x07e, 0x07c, 0x078, 0x260, 0x360, 0x360, 0x1e0, 0x1e0, 0x0e0, 0x0f0, 0x070, 0x070}};
    return t[bank][s];
}
__host__ __device__ constexpr bool wide_has(int bank, int s, int g) { return (wide_mask(bank, s) >> g) & 1u; }
__host__ __device__ constexpr int popc16(unsigned v) {
    int n = 0;
    for (int i = 0; i < 16; ++i) n += (v >> i) & 1u;
    return n;
}
// index of pair (s, g) in its bank's fragment image (pairs in slot order, groups ascending inside a slot)
__host__ __device__ constexpr int wide_pair(int bank, int s, int g) {
    int n = 0;
    for (int k = 0; k < s; ++k) n += popc16(wide_mask(bank, k));
    return n + popc16(wide_mask(bank, s) & ((1u << g) - 1u));
}
__host__ __device__ constexpr int wide_pairs(int bank) { return wide_pair(bank, NSLOT - 1, 16); }
static_assert(wide_pairs(0) == 16 && wide_pairs(1) == 62, "the 80-bin banded tables");
// which table a packed bank's banded image (FBQ) is laid out for: 0 = slot_group (banks of <= 40 columns of any other
// filterbank), 1 / 2 = the lower / upper bank of an 80-bin filterbank (wide_mask)
__host__ __device__ constexpr bool table_has(int table, int s, int g) { return table == 0 ? slot_has_group(s, g) : wide_has(table - 1, s, g); }

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
// (a device function since round 5: the kernel below in frontend.hip is its wrapper; `bidx` of `nblk` workgroups share the quads)
// WIDE (round 6): NUM_MELS = 80 in ONE pass over the spectrum -- the power values of a quad stay in registers while both banks'
// banded contractions run (four passes of five mel groups: 78 MFMAs per quad against 2 x 170 in two launches that each repeated
// the transform); `fbp` is then the two-bank packed buffer and M = 80.
template <int NWAVES, int NGRP, bool WIDE = false>
__device__ __forceinline__ void logmel_body(const float* __restrict__ pcm, int L, long ld, int T, int total_frames,
                                            const float* __restrict__ fbp, int M, float log_eps, const float* __restrict__ zmuv,
                                            float* __restrict__ out, int layout, int n_quads, int aligned, unsigned bidx, unsigned nblk,
                                            int Mo /* mel bins per frame of `out` (= M, or the whole filterbank when this launch is one bank of it) */) {
    constexpr int XR = FeGeom<NWAVES>::XR, XF = FeGeom<NWAVES>::XF;
    __shared__ v2f xch[NWAVES * QUAD * XF];             // FFT transpose tiles, private per wave
    __shared__ v4f c_tab[HOWL_FE_CONST_FLOATS / 4];     // window | W_256 | W_512 rows (read-only after the prologue)
    constexpr int NPAIR_W = 16 + 62;                     // wide_pairs(0) + wide_pairs(1)
    __shared__ v4f c_frag[WIDE ? NPAIR_W * 16 : NSLOT * 64];   // banded filterbank fragments ([slot][lane][4] | WIDE: [pair][lane])
    static_assert(!WIDE || NGRP == NG_BANDED, "the two-bank form runs its banks as passes of five groups");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i1 = lane >> 4, n2 = lane & 15;              // FFT step 1: frame i1 of the quad, column n2
    const int j = lane >> 2, i2 = lane & 3;                // FFT step 2 and everything after: bin class j, frame i2

    // this workgroup's contiguous share of the quads, dealt round-robin to its waves
    const int q_lo = (int)(((long)n_quads * bidx) / nblk);
    const int q_hi = (int)(((long)n_quads * (bidx + 1)) / nblk);
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
        if constexpr (WIDE) {      // bank 0's 16 pairs, then bank 1's 62
            const v4f* f0 = reinterpret_cast<const v4f*>(fbp + FBQ_OFF);
            const v4f* f1 = reinterpret_cast<const v4f*>(fbp + HOWL_FB_PACKED_FLOATS + FBQ_OFF);
            for (int i = tid; i < NPAIR_W * 16; i += NWAVES * 64) c_frag[i] = i < 16 * 16 ? f0[i] : f1[i - 16 * 16];
        } else {
            const v4f* fq = reinterpret_cast<const v4f*>(fbp + FBQ_OFF);
            for (int i = tid; i < NSLOT * 64; i += NWAVES * 64) c_frag[i] = fq[i];
        }
    }
    const bool banded = WIDE ? (reinterpret_cast<const int*>(fbp + FBF_OFF)[1] != 0 &&
                                reinterpret_cast<const int*>(fbp + HOWL_FB_PACKED_FLOATS + FBF_OFF)[1] != 0)
                             : (NGRP == NG_BANDED && reinterpret_cast<const int*>(fbp + FBF_OFF)[0] != 0);   // wave-uniform
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
            o_base = (long)g_frame * Mo;
            o_ms = 1;
        } else {
            int t = t0 + r_out, b = b0;
            if (t >= T) { t -= T; ++b; }
            if (t >= T) { t -= T; ++b; }
            o_base = (long)b * Mo * T + t;
            o_ms = T;
        }
        auto mel_pass = [&](auto g0c, auto bankc) {
            constexpr int G0 = decltype(g0c)::value;
            constexpr int BK = decltype(bankc)::value;         // WIDE: which bank of the 80-bin filterbank this pass contracts
            const int Mb = WIDE ? WIDE_BANK : M;
            f32x4 acc[NH];
#pragma unroll
            for (int g = 0; g < NH; ++g) acc[g] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (WIDE && banded) {
                const float* fw = reinterpret_cast<const float*>(c_frag) + (BK == 0 ? 0 : 16 * 64);
#pragma unroll
                for (int s = 0; s < NSLOT; ++s) {
#pragma unroll
                    for (int g = G0; g < G0 + NH; ++g)
                        if (wide_has(BK, s, g))                    // compile-time after unrolling
                            acc[g - G0] = __builtin_amdgcn_mfma_f32_4x4x1f32(P[s], fw[wide_pair(BK, s, g) * 64 + frow], acc[g - G0], 0, 0, 0);
                }
            } else if (banded) {
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
                const float* fdense = fbp + (BK == 0 ? 0 : HOWL_FB_PACKED_FLOATS) + FBD_OFF;
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
                if (w < ND && u < NH && m < Mb && g_frame < total_frames) {
                    float y = __builtin_amdgcn_logf(vd[h] + log_eps) * 0.69314718055994530942f;
                    y = (y - zm_mean) * zm_rstd;
                    out[o_base + (long)(m + WIDE_BANK * BK) * o_ms] = y;
                }
            }
        };
        mel_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        HOWL_FE_PROBE(wave, lane, pslot++);   // contracted (first half)
        mel_pass(std::integral_constant<int, NH>{}, std::integral_constant<int, 0>{});
        if constexpr (WIDE) {
            mel_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            mel_pass(std::integral_constant<int, NH>{}, std::integral_constant<int, 1>{});
        }
        HOWL_FE_PROBE(wave, lane, pslot++);   // stored
        if (has_next) apply_window(xn);
        b0 = bn;
        t0 = tn;
    }
}


// What a launch of logmel_body needs, checked and derived from howl_logmel_fwd's arguments (one place for the entry point and
// for callers that run the body as rider blocks of another launch).
struct LogmelLaunch {
    const float* pcm;
    int L;
    long ld;
    int T, total;
    const float* fbp;
    int M;
    float log_eps;
    const float* zmuv;
    float* out;
    int layout, n_quads, aligned;
    int Mo;      // mel bins per frame of `out`: M, or the whole filterbank's when the launch computes one bank of it
};
// More than HOWL_FB_COLS mel bins: two banks [0, lo) and [lo, M), lo a multiple of 4 (the contraction's group width)
inline int fb_banks(int M) { return M <= HOWL_FB_COLS ? 1 : 2; }
inline int fb_bank_lo(int M) { return M <= HOWL_FB_COLS ? M : 4 * ((M + 7) / 8); }
inline int logmel_prepare(const float* pcm, int B, int L, long ld, const float* fbp, int M, float log_eps, const float* zmuv,
                          float* out, int layout, LogmelLaunch* ll) {
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
    // 8-byte sample loads need even row strides and an 8-byte aligned base; anything else takes the per-sample path
    const int aligned = ((ld & 1) == 0 && (reinterpret_cast<uintptr_t>(pcm) & 7) == 0) ? 1 : 0;
    *ll = LogmelLaunch{pcm, L, ld, T, total, fbp, M, log_eps, zmuv, out, layout, (total + QUAD - 1) / QUAD, aligned, M};
    return HOWL_OK;
}

}  // namespace
