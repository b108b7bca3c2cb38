// Audio frontend kernels for gfx950 (MI355X): fused windowed-FFT + mel + log (K1), deltas (K2),
// ZMUV statistics (K4), SpecAugment masks (K5), mel-filterbank packing / VTLP construction (K3).
//
// Replaces the ATen/torchaudio op chain the reference runs at
//   howl/data/transform/transform.py:249-254,271-280  (MelSpectrogram -> +1e-7 -> log -> ComputeDeltas x2)
//   howl/data/transform/transform.py:373-410,429-449  (VTLP filterbank)
//   howl/data/transform/operator.py:119-146           (ZmuvTransform)
//   howl/data/transform/transform.py:299-339          (SpecAugmentTransform masks)
//
// K1 design (one launch for the whole batch):
//  * frames are flattened over (utterance, t): g = b*T + t; a workgroup (4 waves) owns 16 consecutive frames
//    per iteration and is persistent over chunks (grid-stride), so its filterbank fragments, Hann window and
//    twiddles stay in registers;
//  * PCM is read straight from HBM/L2 as 256-B coalesced rows (lane j reads sample 64*n1 + j), with the
//    reflect padding of torch.stft(center=True) folded into the index;
//  * two real frames ride in one complex FFT-512 (frame A = re, frame B = im). The FFT is radix-8 x 3 on one
//    wavefront: 8 points per lane in registers, two transposes through a private LDS scratch;
//  * |X|^2 for 257 bins lands in LDS as a [16][260] tile, and the dense mel contraction runs on
//    v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain): M = 16 frames, N = 16*NT mel bins, K = 260 split
//    over the 4 waves, partials combined through LDS;
//  * epilogue: log(x + eps), optional ZMUV (x - mean) / std, store as (B,T,M) [model layout] or (B,M,T).
// Algorithmic HBM bytes per utterance: 4*L read + 4*M*T written (76,960 B at L=16000, M=40).
#include "howl_common.hip.h"
#include "howl_tables.h"
#include "../../include/howl_hip.h"

namespace {

constexpr int N_FFT = 512;
constexpr int HOP = 200;
constexpr int N_FREQ = 257;
constexpr int K_PAD = 260;          // 257 padded to a multiple of the MFMA K (4)
constexpr int K_STEPS = K_PAD / 4;  // 65
constexpr int P_STRIDE = 261;       // LDS row stride of the power tile (odd: spreads frames over banks)
constexpr int X1_STRIDE = 68;       // exchange-1 row stride (complex elements), conflict-free
constexpr int X2_STRIDE = 66;       // exchange-2 row stride
constexpr int SCR_CF = 8 * X1_STRIDE;  // complex elements of scratch per wave (544 >= 512)
constexpr int CHUNK = 16;           // frames per workgroup iteration == MFMA M

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

template <int NT>
__global__ __launch_bounds__(256) void logmel_kernel(const float* __restrict__ pcm, int L, long ld, int T,
                                                     long total_frames, const float* __restrict__ fbp, int M,
                                                     float log_eps, const float* __restrict__ zmuv,
                                                     float* __restrict__ out, int layout, int n_chunks) {
    constexpr int NCOL = 16 * NT;
    __shared__ float P[CHUNK * P_STRIDE];
    __shared__ cf scratch[4 * SCR_CF];  // FFT transposes; reused as the 4 x [16][NCOL] partial-sum tiles
    static_assert(4 * SCR_CF * 2 >= 4 * CHUNK * NCOL, "partial tiles must fit in the FFT scratch");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    cf* scr = scratch + wave * SCR_CF;

    // ---- per-lane constants, resident for the whole persistent loop ----------------------------------
    float win[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) win[n1] = HOWL_HANN512[64 * n1 + lane];
    float tw1r[8], tw1i[8], tw2r[8], tw2i[8];
    const int b_of_lane = lane >> 3;  // stage-2 ownership: lane = k1 + 8*b
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        tw1r[k] = HOWL_TW512[(lane * 8 + k) * 2];
        tw1i[k] = HOWL_TW512[(lane * 8 + k) * 2 + 1];
        tw2r[k] = HOWL_TW64[(b_of_lane * 8 + k) * 2];
        tw2i[k] = HOWL_TW64[(b_of_lane * 8 + k) * 2 + 1];
    }
    // this wave's K slice of the mel contraction: wave 0 -> k-steps [0,17), wave w -> [17+16(w-1), +16)
    const int ks0 = (wave == 0) ? 0 : 17 + 16 * (wave - 1);
    const int nks = (wave == 0) ? 17 : 16;
    float bfrag[17][NT];
#pragma unroll
    for (int i = 0; i < 17; ++i) {
        const int krow = 4 * (ks0 + i) + (lane >> 4);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            bfrag[i][nt] = (i < nks) ? fbp[(long)krow * NCOL + 16 * nt + (lane & 15)] : 0.0f;
    }
    float zm_mean = 0.0f, zm_std = 1.0f;
    if (zmuv != nullptr) {
        zm_mean = zmuv[0];
        zm_std = zmuv[1];
    }

    // Raw samples of one frame pair (this wave's pair `pr` of chunk `chunk`): reflect-padded centre framing.  They are
    // requested one pair ahead of the FFT that consumes them, so the HBM round trip is spent under the previous pair's
    // butterflies instead of in front of every FFT.
    auto fetch_pair = [&](int chunk, int pr, float (&xa)[8], float (&xb)[8]) {
        const long ga = (long)chunk * CHUNK + 4 * wave + 2 * pr, gb = ga + 1;
        const bool va = chunk < n_chunks && ga < total_frames, vb = chunk < n_chunks && gb < total_frames;
        const long ba = va ? ga / T : 0, bb = vb ? gb / T : 0;
        const int ta = va ? (int)(ga - ba * T) : 0, tb = vb ? (int)(gb - bb * T) : 0;
        const float* rowa = pcm + ba * ld;
        const float* rowb = pcm + bb * ld;
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) {
            const int n = 64 * n1 + lane;
            int sa = HOP * ta - N_FFT / 2 + n;
            int sb = HOP * tb - N_FFT / 2 + n;
            sa = sa < 0 ? -sa : sa;
            sb = sb < 0 ? -sb : sb;
            sa = sa >= L ? 2 * (L - 1) - sa : sa;
            sb = sb >= L ? 2 * (L - 1) - sb : sb;
#if defined(HOWL_DIAG_LOGMEL_NOLOAD)   // diagnostic build (tools/variants.py): no PCM reads
            const float a = 1e-3f * (float)sa, b = 1e-3f * (float)sb;
#else
            const float a = rowa[sa], b = rowb[sb];  // always in range (frame 0 of row 0 for invalid frames)
#endif
            xa[n1] = va ? a : 0.0f;
            xb[n1] = vb ? b : 0.0f;
        }
    };
    float xa[8], xb[8], na[8], nb[8];
    fetch_pair(blockIdx.x, 0, xa, xb);

    for (int chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const long g0 = (long)chunk * CHUNK;
        // ---- FFT phase: this wave transforms frame pairs (4w, 4w+1) and (4w+2, 4w+3) -----------------
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int fa = 4 * wave + 2 * pr;  // frame slot of the "real" frame; the "imag" frame is fa + 1
            if (pr == 0) fetch_pair(chunk, 1, na, nb);
            else fetch_pair(chunk + (int)gridDim.x, 0, na, nb);
            cf v[8];
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) v[n1] = {xa[n1] * win[n1], xb[n1] * win[n1]};
#if !defined(HOWL_DIAG_LOGMEL_NOFFT)   // diagnostic build: butterflies and LDS exchanges removed
            // stage 1: radix-8 over n1 (stride 64), twiddle W_512^(lane*k1)
            dft8(v);
#pragma unroll
            for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], tw1r[k], tw1i[k]);
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
#endif
            wave_lds_sync();
#pragma unroll
            for (int d = 0; d < 8; ++d) scr[lane + 64 * d] = v[d];
            wave_lds_sync();
            // separate the two real spectra and take |X|^2 for bins 0..256 (lane 0 also does bin 256)
            float* Pa = P + fa * P_STRIDE;
            float* Pb = Pa + P_STRIDE;
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
                } else if (lane < 4) {  // zero the K padding (bins 257..259)
                    Pa[k] = 0.0f;
                    Pb[k] = 0.0f;
                }
            }
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) {
                xa[n1] = na[n1];
                xb[n1] = nb[n1];
            }
        }
        __syncthreads();
        // ---- mel contraction on the matrix cores: D[frame][mel] += P[frame][k] * fb[k][mel] ------------
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = {0.0f, 0.0f, 0.0f, 0.0f};
        {
            const float* arow = P + (lane & 15) * P_STRIDE + 4 * ks0 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < 17; ++i) {
#if defined(HOWL_DIAG_LOGMEL_NOMEL)   // diagnostic build: mel contraction removed
                if (i == 0) acc[0][0] = arow[0];
#else
                if (i < nks) {
                    const float a = arow[4 * i];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bfrag[i][nt], acc[nt], 0, 0, 0);
                }
#endif
            }
        }
        float* part = reinterpret_cast<float*>(scratch);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                part[(wave * CHUNK + (lane >> 4) * 4 + r) * NCOL + 16 * nt + (lane & 15)] = acc[nt][r];
        __syncthreads();
        // ---- epilogue: combine the 4 K-slices, log, ZMUV, store ---------------------------------------
        for (int idx = tid; idx < CHUNK * NCOL; idx += 256) {
            const int f = idx / NCOL, m = idx - f * NCOL;
            const long g = g0 + f;
            if (m < M && g < total_frames) {
                float s = part[(0 * CHUNK + f) * NCOL + m] + part[(1 * CHUNK + f) * NCOL + m];
                s += part[(2 * CHUNK + f) * NCOL + m];
                s += part[(3 * CHUNK + f) * NCOL + m];
                float y = logf(s + log_eps);
                if (zmuv != nullptr) y = (y - zm_mean) / zm_std;
                if (layout == 1) {
                    out[g * M + m] = y;
                } else {
                    const long b = g / T;
                    const int t = (int)(g - b * T);
                    out[(b * M + m) * T + t] = y;
                }
            }
        }
        __syncthreads();
    }
}

// fb (257, M) row-major -> fbp (260, NCOL) zero padded
__global__ void fb_pack_kernel(const float* __restrict__ fb, int M, float* __restrict__ fbp, int ncol) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= K_PAD * ncol) return;
    const int k = idx / ncol, m = idx - k * ncol;
    fbp[idx] = (k < N_FREQ && m < M) ? fb[k * M + m] : 0.0f;
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

// running update of operator.py:133-135 on the device buffers (no host sync)
__global__ void zmuv_update_kernel(const double* __restrict__ sums, double count, float* total, float* mean, float* mean2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
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
    HOWL_CHECK_LAUNCH("howl_fb_pack");
    return HOWL_OK;
}

int howl_fb_from_points(const HowlMelPoints* pts, int M, float nyquist, float* fbp, hipStream_t stream) {
    HOWL_REQUIRE(pts && fbp, "howl_fb_from_points: null pointer");
    HOWL_REQUIRE(M >= 1 && M <= HOWL_MAX_MELS, "howl_fb_from_points: M=%d unsupported (1..%d)", M, HOWL_MAX_MELS);
    const int ncol = HOWL_FB_COLS;
    hipLaunchKernelGGL(fb_points_kernel, dim3((K_PAD * ncol + 255) / 256), dim3(256), 0, stream, *pts, M, nyquist, fbp,
                       ncol);
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
    const long total = (long)B * T;
    const int n_chunks = (int)((total + CHUNK - 1) / CHUNK);
    int grid = howl_num_cus() * 2;   // two workgroups are resident per CU (206 VGPRs): each pays the constant prologue once
    if (grid > n_chunks) grid = n_chunks;
    {
        HowlProfScope prof("logmel", stream);
        hipLaunchKernelGGL(logmel_kernel<HOWL_FB_COLS / 16>, dim3(grid), dim3(256), 0, stream, pcm, L, ld, T, total, fbp, M,
                           log_eps, zmuv, out, layout, n_chunks);
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
