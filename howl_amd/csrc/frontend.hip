// Audio frontend kernels for gfx950 (MI355X): fused windowed-FFT + mel + log (K1), deltas (K2),
// ZMUV statistics (K4), SpecAugment masks (K5), mel-filterbank packing / VTLP construction (K3).
//
// Replaces the ATen/torchaudio op chain the reference runs at
//   howl/data/transform/transform.py:249-254,271-280  (MelSpectrogram -> +1e-7 -> log -> ComputeDeltas x2)
//   howl/data/transform/transform.py:373-410,429-449  (VTLP filterbank)
//   howl/data/transform/operator.py:119-146           (ZmuvTransform)
//   howl/data/transform/transform.py:299-339          (SpecAugmentTransform masks)
//
// K1 design (one launch for the whole batch): every WAVEFRONT is an independent worker -- no workgroup barrier anywhere.
//  * frames are flattened over (utterance, t): g = b*T + t; a wave owns "quads" of 4 consecutive frames (grid-stride), i.e.
//    two frame PAIRS, and keeps its Hann window and twiddles in registers;
//  * PCM is read straight from HBM/L2 as 256-B coalesced rows (lane j reads sample 64*n1 + j), with the reflect padding of
//    torch.stft(center=True) folded into the index; the next pair's samples are requested before the current FFT starts;
//  * two real frames ride in one complex FFT-512 (frame A = re, frame B = im): radix-8 x 3 on one wavefront, 8 points per
//    lane in registers, two transposes through the wave's private LDS scratch (wave-scope syncs only);
//  * |X|^2 for bins 0..256 of the quad's 4 frames lands in the wave's private [4][296] LDS tile, and the mel contraction
//    runs on v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products per instruction, exact fp32 FMA chains): block j
//    takes bin 16*kg + j, its 4 rows are the 4 frames, its 4 columns one group of 4 mel bins -- no padding of the frame
//    dimension to a 16-row tile, so a quad needs ~56 instructions for the standard banded filterbank.  The B operands come
//    from a fragment-ordered copy of the filterbank (built once by howl_fb_pack / howl_fb_from_points together with the
//    band limits [lo, hi) of every mel group) through L1; the 16 blocks are summed by a 15-shuffle reduce-scatter that
//    leaves lane 16*frame + mel%16 holding one output value;
//  * epilogue: log(x + eps), optional ZMUV (x - mean) / std, 64-B contiguous stores as (B,T,M) [model layout] or (B,M,T).
// 9 KB of LDS and <= 128 VGPRs per wave: 16 waves per CU hide the LDS / L2 latencies of the serial FFT stages.
// Algorithmic HBM bytes per utterance: 4*L read + 4*M*T written (76,960 B at L=16000, M=40).
#include <stdlib.h>

#include "howl_common.hip.h"
#include "howl_tables.h"
#include "../../include/howl_hip.h"

namespace {

constexpr int N_FFT = 512;
constexpr int HOP = 200;
constexpr int N_FREQ = 257;
constexpr int K_PAD = 260;          // rows of the row-major packed filterbank (257 padded to a multiple of 4)
constexpr int X1_STRIDE = 68;       // exchange-1 row stride (complex elements), conflict-free
constexpr int X2_STRIDE = 66;       // exchange-2 row stride
constexpr int SCR_CF = 8 * X1_STRIDE;  // complex elements of scratch per wave (544 >= 512)
constexpr int QUAD = 4;             // frames per wave iteration == rows of a 4x4x1 MFMA block
constexpr int KG = 17;              // bin groups of 16 (one bin per MFMA block): 17 * 16 = 272 >= 257
constexpr int NG = HOWL_FB_COLS / 4;   // mel groups of 4 (columns of a block): 12
constexpr int PQ_STRIDE = 296;      // row stride of the power tile: >= 272 and = 8 (mod 32), so the A-operand read
                                    // (lane 4j+i -> row i, bin 16kg + j) touches 32 distinct banks per half wave
// packed filterbank buffer: [ (260, 48) row-major | KG x NG fragments of 64 lanes | 2 * NG band limits (int32) | pad ]
constexpr int FBQ_OFF = K_PAD * HOWL_FB_COLS;
constexpr int FBQ_BAND_OFF = FBQ_OFF + KG * NG * 64;
static_assert(FBQ_BAND_OFF + 32 == HOWL_FB_PACKED_FLOATS, "include/howl_hip.h and the kernels disagree on the packed filterbank size");

struct cf {
    float re, im;
};

__device__ __forceinline__ cf cmul(cf a, float wr, float wi) { return {a.re * wr - a.im * wi, a.re * wi + a.im * wr}; }
__device__ __forceinline__ cf cadd(cf a, cf b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return {a.re - b.re, a.im - b.im}; }

// 4-point DFT of (a0..a3), forward sign; results in natural order o0..o3
__device__ __forceinline__ void dft4(cf a0, cf a1, cf a2, cf a3, cf& o0, cf& o1, cf& o2, cf& o3) {
    cf b0 = cadd(a0, a2), b1 = cadd(a1, a3), b2 = csub(a0, a2), d = csub(a1, a3);
    cf b3 = {d.im, -d.re};  // (a1 - a3) * (-i)
    o0 = cadd(b0, b1);
    o2 = csub(b0, b1);
    o1 = cadd(b2, b3);
    o3 = csub(b2, b3);
}

// in-place 8-point forward DFT: v[k] = sum_n v[n] * exp(-2 pi i n k / 8)
__device__ __forceinline__ void dft8(cf (&v)[8]) {
    const float h = 0.70710678118654752440f;
    cf t0 = cadd(v[0], v[4]), t1 = cadd(v[1], v[5]), t2 = cadd(v[2], v[6]), t3 = cadd(v[3], v[7]);
    cf d0 = csub(v[0], v[4]), e1 = csub(v[1], v[5]), e2 = csub(v[2], v[6]), e3 = csub(v[3], v[7]);
    cf d1 = {(e1.re + e1.im) * h, (e1.im - e1.re) * h};   // * W8^1 = (1 - i)/sqrt2
    cf d2 = {e2.im, -e2.re};                              // * W8^2 = -i
    cf d3 = {(e3.im - e3.re) * h, -(e3.re + e3.im) * h};  // * W8^3 = (-1 - i)/sqrt2
    dft4(t0, t1, t2, t3, v[0], v[2], v[4], v[6]);
    dft4(d0, d1, d2, d3, v[1], v[3], v[5], v[7]);
}

// One butterfly step of the 16-block reduce-scatter: lanes whose bit `XOR` is set keep the upper half of v[0..2n), the
// others the lower half; each adds what its partner (lane ^ XOR) gives up.
template <int N, int XOR>
__device__ __forceinline__ void reduce_scatter_step(float (&v)[16], bool upper) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float send = upper ? v[i] : v[i + N];
        const float keep = upper ? v[i + N] : v[i];
        v[i] = keep + __shfl_xor(send, XOR);
    }
}

#if defined(HOWL_DIAG_PROBE)  // diagnostic build (tools/probe_logmel.py): s_memtime stamps of workgroup 0, [wave][slot]
__device__ unsigned long long* g_howl_probe_fe = nullptr;
#define HOWL_FE_PROBE(wave_, lane_, slot_)                                                 \
    do {                                                                                   \
        if (g_howl_probe_fe != nullptr && blockIdx.x == 0 && (lane_) == 0 && (slot_) < 64) \
            g_howl_probe_fe[(wave_) * 64 + (slot_)] = __builtin_amdgcn_s_memtime();         \
    } while (0)
#else
#define HOWL_FE_PROBE(wave_, lane_, slot_) ((void)0)
#endif

// A frame of the flattened (utterance, t) sequence; every field is wave-uniform (lives in SGPRs).
struct FrameRef {
    int b, t;
    bool valid;
};

__global__ __launch_bounds__(256, 4) void logmel_kernel(const float* __restrict__ pcm, int L, long ld, int T,
                                                        int total_frames, const float* __restrict__ fbp, int M,
                                                        float log_eps, const float* __restrict__ zmuv,
                                                        float* __restrict__ out, int layout, int n_quads) {
    __shared__ cf scratch[4 * SCR_CF];            // FFT transposes, private per wave
    __shared__ float Pq[4 * QUAD * PQ_STRIDE];    // power tile, private per wave
    __shared__ cf tw1[7 * 64];                    // stage-1 twiddles W_512^(lane*k), k = 1..7 (read-only after the prologue)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    cf* scr = scratch + wave * SCR_CF;
    float* P = Pq + wave * QUAD * PQ_STRIDE;

    // ---- per-lane constants, resident for the whole persistent loop ----------------------------------
    float win[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) win[n1] = HOWL_HANN512[64 * n1 + lane];
    float tw2r[8], tw2i[8];
    const int b_of_lane = lane >> 3;  // stage-2 ownership: lane = k1 + 8*b
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        tw2r[k] = HOWL_TW64[(b_of_lane * 8 + k) * 2];
        tw2i[k] = HOWL_TW64[(b_of_lane * 8 + k) * 2 + 1];
    }
    for (int i = tid; i < 7 * 64; i += 256) {     // [k-1][lane]: a wave reads 64 consecutive entries per k
        const int k = 1 + i / 64, l = i & 63;
        tw1[i] = {HOWL_TW512[(l * 8 + k) * 2], HOWL_TW512[(l * 8 + k) * 2 + 1]};
    }
    float zm_mean = 0.0f, zm_std = 1.0f;
    if (zmuv != nullptr) {
        zm_mean = zmuv[0];
        zm_std = zmuv[1];
    }
    // bins 257..271 of the tile are read by the last bin group (times a zero filterbank entry): keep them zero
    if (lane < QUAD * 16) P[(lane >> 4) * PQ_STRIDE + 256 + (lane & 15)] = 0.0f;
    __syncthreads();                              // the only workgroup barrier of the kernel (tw1 filled)
    const int* band = reinterpret_cast<const int*>(fbp + FBQ_BAND_OFF);
    const float* frag = fbp + FBQ_OFF + lane;
    const int n_groups = (M + 3) >> 2;

    // frame g of the flattened sequence -> (utterance, t): one scalar division per quad, the other frames by stepping
    auto frame_at = [&](int g) {
        FrameRef f;
        f.valid = g < total_frames;
        const int gg = f.valid ? g : 0;
        f.b = gg / T;
        f.t = gg - f.b * T;
        return f;
    };
    auto next_frame = [&](FrameRef f, int g) {    // g = index of the frame after f
        ++f.t;
        if (f.t >= T) {
            f.t = 0;
            ++f.b;
        }
        f.valid = g < total_frames;
        if (!f.valid) f.b = f.t = 0;
        return f;
    };
    // Raw samples of one frame, centre framing with reflect padding (torch.stft(center=True)).  Interior frames (all but
    // the first two and last two or three of an utterance) are eight coalesced 256-B rows at immediate offsets from one
    // wave-uniform base; only edge frames pay for the per-sample index arithmetic.  Invalid frames read nothing.
    auto fetch_frame = [&](const FrameRef& f, float (&x)[8]) {
        const float* row = pcm + (long)f.b * ld;
        const int s0 = HOP * f.t - N_FFT / 2;
        if (!f.valid) {
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) x[n1] = 0.0f;
        } else if (s0 >= 0 && s0 + N_FFT <= L) {
            const float* p0 = row + s0 + lane;
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) x[n1] = p0[64 * n1];
        } else {
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) {
                int sa = s0 + 64 * n1 + lane;
                sa = sa < 0 ? -sa : sa;
                sa = sa >= L ? 2 * (L - 1) - sa : sa;
                x[n1] = row[sa];
            }
        }
    };
    // Two real frames (xa -> real part, xb -> imaginary part) through one complex FFT-512; |X|^2 of both to rows Pa, Pb.
    auto fft_pair = [&](const float (&xa)[8], const float (&xb)[8], float* Pa, float* Pb) {
        cf v[8];
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) v[n1] = {xa[n1] * win[n1], xb[n1] * win[n1]};
        // stage 1: radix-8 over n1 (stride 64), twiddle W_512^(lane*k1)
        dft8(v);
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            const cf w = tw1[(k - 1) * 64 + lane];
            v[k] = cmul(v[k], w.re, w.im);
        }
        wave_lds_sync();  // previous users of this scratch (power phase of the last pair) are done
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) scr[k1 * X1_STRIDE + lane] = v[k1];
        wave_lds_sync();
        {   // stage 2: lane owns (k1 = lane & 7, b = lane >> 3), radix-8 over a with n2 = 8a + b
            const int k1 = lane & 7, b = lane >> 3;
#pragma unroll
            for (int a = 0; a < 8; ++a) v[a] = scr[k1 * X1_STRIDE + 8 * a + b];
            dft8(v);
#pragma unroll
            for (int c = 1; c < 8; ++c) v[c] = cmul(v[c], tw2r[c], tw2i[c]);
            wave_lds_sync();
#pragma unroll
            for (int c = 0; c < 8; ++c) scr[k1 * X2_STRIDE + c * 8 + b] = v[c];
            wave_lds_sync();
        }
        {   // stage 3: lane owns (k1 = lane & 7, c = lane >> 3), radix-8 over b -> Z[lane + 64 d]
            const int k1 = lane & 7, c = lane >> 3;
#pragma unroll
            for (int b = 0; b < 8; ++b) v[b] = scr[k1 * X2_STRIDE + c * 8 + b];
            dft8(v);
        }
        wave_lds_sync();
#pragma unroll
        for (int d = 0; d < 8; ++d) scr[lane + 64 * d] = v[d];
        wave_lds_sync();
        // separate the two real spectra and take |X|^2 for bins 0..256 (lane 0 also does bin 256)
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const int k = lane + 64 * r;
            if (r < 4 || lane == 0) {
                const cf zk = (r < 4) ? v[r] : v[4];
                const cf zn = scr[(N_FFT - k) & (N_FFT - 1)];
                const float are = 0.5f * (zk.re + zn.re), aim = 0.5f * (zk.im - zn.im);
                const float bre = 0.5f * (zk.im + zn.im), bim = -0.5f * (zk.re - zn.re);
                Pa[k] = are * are + aim * aim;
                Pb[k] = bre * bre + bim * bim;
            }
        }
    };

    const int wave_id = (int)blockIdx.x * 4 + wave, n_waves = (int)gridDim.x * 4;
    int pslot = 0;
    HOWL_FE_PROBE(wave, lane, pslot++);   // prologue done
    float xa[8], xb[8], ya[8], yb[8];
    FrameRef f0 = frame_at(QUAD * wave_id);
    FrameRef f1 = next_frame(f0, QUAD * wave_id + 1);
    if (wave_id < n_quads) {
        fetch_frame(f0, xa);
        fetch_frame(f1, xb);
    }

    for (int quad = wave_id; quad < n_quads; quad += n_waves) {
        const int g0 = QUAD * quad;
        // ---- FFT phase: frame pairs (g0, g0+1) and (g0+2, g0+3); the next pair's samples are requested first ---------
        const FrameRef f2 = next_frame(f1, g0 + 2), f3 = next_frame(f2, g0 + 3);
        HOWL_FE_PROBE(wave, lane, pslot++);   // quad start
        fetch_frame(f2, ya);
        fetch_frame(f3, yb);
        fft_pair(xa, xb, P, P + PQ_STRIDE);
        HOWL_FE_PROBE(wave, lane, pslot++);   // first pair transformed
        const FrameRef q0 = f0, q1 = f1;          // this quad's frames, for the epilogue
        const int gn = QUAD * (quad + n_waves);   // first frame of this wave's next quad (invalid beyond the batch)
        f0 = frame_at(gn);
        f1 = next_frame(f0, gn + 1);
        fetch_frame(f0, xa);
        fetch_frame(f1, xb);
        fft_pair(ya, yb, P + 2 * PQ_STRIDE, P + 3 * PQ_STRIDE);
        wave_lds_sync();
        HOWL_FE_PROBE(wave, lane, pslot++);   // second pair transformed
        // ---- mel contraction: D_j[frame][mel] += P[frame][16 kg + j] * fb[16 kg + j][mel], blocks j summed afterwards ----
        const float* arow = P + (lane & 3) * PQ_STRIDE + (lane >> 2);
        // this lane's output after the reduce-scatter: frame r_out of the quad, mel column 16 * batch + c_out
        const int r_out = lane >> 4, c_out = lane & 15;
        const int g_out = g0 + r_out;
        long o_base;      // element offset of (frame g_out, mel 0); mel stride o_ms
        long o_ms;
        if (layout == 1) {
            o_base = (long)g_out * M;
            o_ms = 1;
        } else {
            const int b_out = r_out == 0 ? q0.b : (r_out == 1 ? q1.b : (r_out == 2 ? f2.b : f3.b));
            const int t_out = r_out == 0 ? q0.t : (r_out == 1 ? q1.t : (r_out == 2 ? f2.t : f3.t));
            o_base = (long)b_out * M * T + t_out;
            o_ms = T;
        }
#pragma unroll
        for (int batch = 0; batch < NG / 4; ++batch) {
            if (4 * batch >= n_groups) break;
            int klo = KG, khi = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int lo = band[2 * (4 * batch + c)], hi = band[2 * (4 * batch + c) + 1];
                if (hi > lo) {
                    klo = lo < klo ? lo : klo;
                    khi = hi > khi ? hi : khi;
                }
            }
            f32x4 acc[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = {0.0f, 0.0f, 0.0f, 0.0f};
            // two bin groups per trip, the second one's operands requested before the first one's MFMAs (a fragment is
            // zero outside its group's band, so every group of the batch runs over the union [klo, khi) unpredicated)
            for (int kg = klo; kg < khi; kg += 2) {
                const bool two = kg + 1 < khi;
                const float* fk = frag + (long)(kg * NG + 4 * batch) * 64;
                const float a0 = arow[16 * kg];
                float b0[4], b1[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) b0[c] = fk[64 * c];
                const float a1 = two ? arow[16 * kg + 16] : 0.0f;
#pragma unroll
                for (int c = 0; c < 4; ++c) b1[c] = two ? fk[NG * 64 + 64 * c] : 0.0f;
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, b0[c], acc[c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, b1[c], acc[c], 0, 0, 0);
            }
            // lane 4j+i holds D_j[r][4c + i] in acc[c][r]: index the 16 values by 4r + c, the block that will own the sum
            float v16[16];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) v16[4 * r + c] = acc[c][r];
            reduce_scatter_step<8, 32>(v16, (lane & 32) != 0);
            reduce_scatter_step<4, 16>(v16, (lane & 16) != 0);
            reduce_scatter_step<2, 8>(v16, (lane & 8) != 0);
            reduce_scatter_step<1, 4>(v16, (lane & 4) != 0);
            // lane 4j+i now holds the full sum for (r, c) = (j >> 2, j & 3), column i: frame r_out, mel 16*batch + c_out
            const int m = 16 * batch + c_out;
            if (m < M && g_out < total_frames) {
                float y = logf(v16[0] + log_eps);
                if (zmuv != nullptr) y = (y - zm_mean) / zm_std;
                out[o_base + (long)m * o_ms] = y;
            }
        }
        HOWL_FE_PROBE(wave, lane, pslot++);   // mel + epilogue done
    }
}

// fb (257, M) row-major -> fbp (260, NCOL) zero padded
__global__ void fb_pack_kernel(const float* __restrict__ fb, int M, float* __restrict__ fbp, int ncol) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= K_PAD * ncol) return;
    const int k = idx / ncol, m = idx - k * ncol;
    fbp[idx] = (k < N_FREQ && m < M) ? fb[k * M + m] : 0.0f;
}

// Second half of a filterbank build (same stream, after the row-major part is written): the fragment-ordered copy read by
// the 4x4x1 MFMAs of logmel_kernel -- fragment (kg, g), lane 4j+c = fb[16 kg + j][4 g + c] -- and, per mel group g, the range
// [lo, hi) of bin groups kg that hold a non-zero weight (lo = hi = 0 for an empty group).  One workgroup: 13 K elements.
__global__ __launch_bounds__(1024) void fb_fragments_kernel(float* __restrict__ fbp) {
    __shared__ int nz[KG * NG];
    const float* rm = fbp;                 // (K_PAD, HOWL_FB_COLS) row-major
    float* fq = fbp + FBQ_OFF;
    for (int p = threadIdx.x; p < KG * NG; p += blockDim.x) nz[p] = 0;
    __syncthreads();
    for (int idx = threadIdx.x; idx < KG * NG * 64; idx += blockDim.x) {
        const int lane = idx & 63, pair = idx >> 6;
        const int kg = pair / NG, g = pair - kg * NG;
        const int k = 16 * kg + (lane >> 2), m = 4 * g + (lane & 3);
        const float v = (k < N_FREQ) ? rm[k * HOWL_FB_COLS + m] : 0.0f;
        fq[idx] = v;
        if (v != 0.0f) nz[pair] = 1;       // benign race: every writer stores 1
    }
    __syncthreads();
    if (threadIdx.x < NG) {
        const int g = threadIdx.x;
        int lo = 0, hi = 0;
        for (int kg = KG - 1; kg >= 0; --kg)
            if (nz[kg * NG + g]) lo = kg;
        for (int kg = 0; kg < KG; ++kg)
            if (nz[kg * NG + g]) hi = kg + 1;
        int* band = reinterpret_cast<int*>(fbp + FBQ_BAND_OFF);
        band[2 * g] = lo;
        band[2 * g + 1] = hi;
    }
}

// triangles from M+2 corner frequencies (already VTLP-warped on the host: 42 scalars), exactly the
// slope arithmetic of transform.py:402-409; all_freqs = linspace(0, sr/2, 257) = k * (sr/2) / 256.
__global__ void fb_points_kernel(HowlMelPoints pts, int M, float nyquist, float* __restrict__ fbp, int ncol) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= K_PAD * ncol) return;
    const int k = idx / ncol, m = idx - k * ncol;
    float v = 0.0f;
    if (k < N_FREQ && m < M) {
        const float f = (k == N_FREQ - 1) ? nyquist : (float)k * (nyquist / (float)(N_FREQ - 1));
        const float down = (-1.0f * (pts.f[m] - f)) / (pts.f[m + 1] - pts.f[m]);
        const float up = (pts.f[m + 2] - f) / (pts.f[m + 2] - pts.f[m + 1]);
        v = fmaxf(0.0f, fminf(down, up));
    }
    fbp[idx] = v;
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
    if (count < 0.0) count = sums[2];
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

#if defined(HOWL_DIAG_PROBE)
extern "C" int howl_diag_set_probe_fe(unsigned long long* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_howl_probe_fe), &buf, sizeof(buf));
}
#endif

extern "C" {

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
    hipLaunchKernelGGL(fb_points_kernel, dim3((K_PAD * ncol + 255) / 256), dim3(256), 0, stream, *pts, M, nyquist, fbp,
                       ncol);
    hipLaunchKernelGGL(fb_fragments_kernel, dim3(1), dim3(1024), 0, stream, fbp);
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
    HOWL_REQUIRE((long)B * T < (1L << 31) - 4 * QUAD * 65536L, "howl_logmel_fwd: B*T = %ld frames exceeds the 32-bit frame index", (long)B * T);
    const int total = B * T;
    const int n_quads = (total + QUAD - 1) / QUAD;
    // four 4-wave workgroups are resident per CU (9 KB of LDS and <= 128 VGPRs per wave); a wave strides over the quads
    int per_cu = 4;
    if (const char* e = getenv("HOWL_LOGMEL_WGS_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) : 4;   // occupancy experiments
    int grid = howl_num_cus() * per_cu;
    if (grid > (n_quads + 3) / 4) grid = (n_quads + 3) / 4;
    {
        HowlProfScope prof("logmel", stream);
        hipLaunchKernelGGL(logmel_kernel, dim3((unsigned)grid), dim3(256), 0, stream, pcm, L, ld, T, total, fbp, M, log_eps, zmuv,
                           out, layout, n_quads);
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

int howl_zmuv_update_masked(const float* x, const float* mask, size_t n, float* total, float* mean, float* mean2,
                            double* scratch3, hipStream_t stream) {
    HOWL_REQUIRE(x && mask && total && mean && mean2 && scratch3, "howl_zmuv_update_masked: null pointer");
    HOWL_REQUIRE(n >= 1, "howl_zmuv_update_masked: empty input");
    hipMemsetAsync(scratch3, 0, 3 * sizeof(double), stream);
    size_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(sum_sumsq_masked_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, mask, n, scratch3);
    hipLaunchKernelGGL(zmuv_update_kernel, dim3(1), dim3(64), 0, stream, scratch3, -1.0, total, mean, mean2);
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
