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

#include "howl_logmel.hip.h"

namespace {

template <int NWAVES, int NGRP, bool WIDE = false>
__global__ __launch_bounds__(NWAVES * 64) void logmel_kernel(const float* __restrict__ pcm, int L, long ld, int T,
                                                             int total_frames, const float* __restrict__ fbp, int M,
                                                             float log_eps, const float* __restrict__ zmuv,
                                                             float* __restrict__ out, int layout, int n_quads, int aligned, int Mo) {
    logmel_body<NWAVES, NGRP, WIDE>(pcm, L, ld, T, total_frames, fbp, M, log_eps, zmuv, out, layout, n_quads, aligned, blockIdx.x, gridDim.x, Mo);
}

// columns [m0, m0 + Mb) of fb (257, M) row-major -> fbp (260, NCOL) zero padded
__global__ void fb_pack_kernel(const float* __restrict__ fb, int M, float* __restrict__ fbp, int ncol, int m0, int Mb) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= K_PAD * ncol) return;
    const int k = idx / ncol, m = idx - k * ncol;
    fbp[idx] = (k < N_FREQ && m < Mb) ? fb[k * M + m0 + m] : 0.0f;
}

// Second half of a filterbank build (same stream, after the row-major part is written): the fragment-ordered copies read by
// the 4x4x1 MFMAs of logmel_kernel -- lane 4j + c of fragment (slot s, group g) = fb[bin_of(s, j)][4 g + c] -- once as the
// banded LDS image ([s][lane][4]: the four groups slot_group(s, .) of a slot side by side) and once for every (s, g) pair,
// plus the flag that tells the kernel whether the banded image covers every non-zero weight.  One workgroup.
// (slot, group) of pair p of a wide table's fragment image (wide_pair's inverse), or slot = -1 past the table's pairs
__device__ __forceinline__ void wide_pair_at(int bank, int p, int& s_out, int& g_out) {
    s_out = -1, g_out = 0;
    int n = 0;
    for (int s = 0; s < NSLOT; ++s)
        for (int g = 0; g < 10; ++g)
            if (wide_has(bank, s, g)) {
                if (n == p) s_out = s, g_out = g;
                ++n;
            }
}
// `table`: 0 = the 40-bin slot_group table; 1 / 2 = the lower / upper bank of the 80-bin filterbank (FBQ pair-major, flag [1])
__global__ __launch_bounds__(1024) void fb_fragments_kernel(float* __restrict__ fbp, int table) {
    __shared__ int uncovered;
    const float* rm = fbp;                 // (K_PAD, HOWL_FB_COLS) row-major
    if (threadIdx.x == 0) uncovered = 0;
    __syncthreads();
    float* fq = fbp + FBQ_OFF;
    for (int idx = threadIdx.x; idx < FBQ_FLOATS; idx += blockDim.x) {
        if (table == 0) {
            const int c = idx & 3, lane = (idx >> 2) & 63, s = idx >> 8;
            const int g = slot_group(s, c), bin = bin_of(s, lane >> 2);
            fq[idx] = (g >= 0 && bin >= 0) ? rm[bin * HOWL_FB_COLS + 4 * g + (lane & 3)] : 0.0f;
        } else {
            const int lane = idx & 63;
            int s, g;
            wide_pair_at(table - 1, idx >> 6, s, g);
            const int bin = s >= 0 ? bin_of(s, lane >> 2) : -1;
            fq[idx] = bin >= 0 ? rm[bin * HOWL_FB_COLS + 4 * g + (lane & 3)] : 0.0f;
        }
    }
    float* fd = fbp + FBD_OFF;
    for (int idx = threadIdx.x; idx < FBD_FLOATS; idx += blockDim.x) {
        const int lane = idx & 63, pair = idx >> 6;
        const int s = pair / NG_MAX, g = pair - s * NG_MAX, bin = bin_of(s, lane >> 2);
        fd[idx] = bin >= 0 ? rm[bin * HOWL_FB_COLS + 4 * g + (lane & 3)] : 0.0f;
    }
    for (int idx = threadIdx.x; idx < N_FREQ * HOWL_FB_COLS; idx += blockDim.x) {
        const int k = idx / HOWL_FB_COLS, m = idx - k * HOWL_FB_COLS;
        if (rm[idx] != 0.0f && !table_has(table, slot_of_bin(k), m >> 2)) uncovered = 1;   // benign race: every writer stores 1
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        reinterpret_cast<int*>(fbp + FBF_OFF)[0] = (table == 0 && !uncovered) ? 1 : 0;
        reinterpret_cast<int*>(fbp + FBF_OFF)[1] = (table != 0 && !uncovered) ? 1 : 0;
    }
}

// triangles from M+2 corner frequencies (already VTLP-warped on the host: 42 scalars), exactly the
// slope arithmetic of transform.py:402-409; all_freqs = linspace(0, sr/2, 257) = k * (sr/2) / 256.
// (bank column m = mel bin m0 + m of the filterbank; M = columns of this bank)
__device__ __forceinline__ float fb_triangle(const HowlMelPoints& pts, int M, float nyquist, int k, int m, int m0) {
    if (k < 0 || k >= N_FREQ || m >= M) return 0.0f;
    const float f = (k == N_FREQ - 1) ? nyquist : (float)k * (nyquist / (float)(N_FREQ - 1));
    const float* pf = pts.f + m0;
    const float down = (-1.0f * (pf[m] - f)) / (pf[m + 1] - pf[m]);
    const float up = (pf[m + 2] - f) / (pf[m + 2] - pf[m + 1]);
    return fmaxf(0.0f, fminf(down, up));
}

// The whole packed filterbank of howl_fb_from_points in ONE launch (round 4; it was the row-major matrix, then a one-workgroup
// kernel gathering the fragment images from it: 5 + 11 us per VTLP step, i.e. on 75 % of the training steps): every element
// of the three images is computed from the corner points where it is stored, the last block takes the coverage flag.
__global__ __launch_bounds__(256) void fb_from_points_kernel(HowlMelPoints pts, int M, float nyquist, float* __restrict__ fbp, int m0,
                                                             int table) {
    constexpr int N_RM = K_PAD * HOWL_FB_COLS;
    if (blockIdx.x == gridDim.x - 1) {      // does the banded image cover every non-zero weight?
        __shared__ int uncovered;
        if (threadIdx.x == 0) uncovered = 0;
        __syncthreads();
        for (int idx = threadIdx.x; idx < N_FREQ * HOWL_FB_COLS; idx += blockDim.x) {
            const int k = idx / HOWL_FB_COLS, m = idx - k * HOWL_FB_COLS;
            if (fb_triangle(pts, M, nyquist, k, m, m0) != 0.0f && !table_has(table, slot_of_bin(k), m >> 2)) uncovered = 1;   // benign race
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            reinterpret_cast<int*>(fbp + FBF_OFF)[0] = (table == 0 && !uncovered) ? 1 : 0;
            reinterpret_cast<int*>(fbp + FBF_OFF)[1] = (table != 0 && !uncovered) ? 1 : 0;
        }
        return;
    }
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < N_RM) {                       // (K_PAD, HOWL_FB_COLS) row-major
        fbp[idx] = fb_triangle(pts, M, nyquist, idx / HOWL_FB_COLS, idx % HOWL_FB_COLS, m0);
        return;
    }
    idx -= N_RM;
    if (idx < FBQ_FLOATS) {                 // banded LDS image: [s][lane][4], or [pair][lane] for the 80-bin banks
        if (table == 0) {
            const int c = idx & 3, lane = (idx >> 2) & 63, sl = idx >> 8;
            const int g = slot_group(sl, c), bin = bin_of(sl, lane >> 2);
            fbp[FBQ_OFF + idx] = (g >= 0 && bin >= 0) ? fb_triangle(pts, M, nyquist, bin, 4 * g + (lane & 3), m0) : 0.0f;
        } else {
            const int lane = idx & 63;
            int sl, g;
            wide_pair_at(table - 1, idx >> 6, sl, g);
            const int bin = sl >= 0 ? bin_of(sl, lane >> 2) : -1;
            fbp[FBQ_OFF + idx] = bin >= 0 ? fb_triangle(pts, M, nyquist, bin, 4 * g + (lane & 3), m0) : 0.0f;
        }
        return;
    }
    idx -= FBQ_FLOATS;
    if (idx < FBD_FLOATS) {                 // every (slot, group) fragment
        const int lane = idx & 63, pair = idx >> 6;
        const int sl = pair / NG_MAX, g = pair - sl * NG_MAX, bin = bin_of(sl, lane >> 2);
        fbp[FBD_OFF + idx] = bin >= 0 ? fb_triangle(pts, M, nyquist, bin, 4 * g + (lane & 3), m0) : 0.0f;
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

size_t howl_fb_packed_floats(int M) { return (size_t)fb_banks(M) * HOWL_FB_PACKED_FLOATS; }

int howl_fb_pack(const float* fb, int M, float* fbp, hipStream_t stream) {
    HOWL_REQUIRE(fb && fbp, "howl_fb_pack: null pointer");
    HOWL_REQUIRE(M >= 1 && M <= HOWL_MAX_MELS, "howl_fb_pack: M=%d unsupported (1..%d)", M, HOWL_MAX_MELS);
    const int ncol = HOWL_FB_COLS;
    for (int bank = 0; bank < fb_banks(M); ++bank) {
        const int m0 = bank == 0 ? 0 : fb_bank_lo(M), Mb = bank == 0 ? fb_bank_lo(M) : M - fb_bank_lo(M);
        float* dst = fbp + (size_t)bank * HOWL_FB_PACKED_FLOATS;
        hipLaunchKernelGGL(fb_pack_kernel, dim3((K_PAD * ncol + 255) / 256), dim3(256), 0, stream, fb, M, dst, ncol, m0, Mb);
        hipLaunchKernelGGL(fb_fragments_kernel, dim3(1), dim3(1024), 0, stream, dst, M == WIDE_MELS ? 1 + bank : 0);
    }
    HOWL_CHECK_LAUNCH("howl_fb_pack");
    return HOWL_OK;
}

int howl_fb_from_points(const HowlMelPoints* pts, int M, float nyquist, float* fbp, hipStream_t stream) {
    HOWL_REQUIRE(pts && fbp, "howl_fb_from_points: null pointer");
    HOWL_REQUIRE(M >= 1 && M <= HOWL_MAX_MELS, "howl_fb_from_points: M=%d unsupported (1..%d)", M, HOWL_MAX_MELS);
    const int ncol = HOWL_FB_COLS;
    (void)ncol;
    const int work = K_PAD * HOWL_FB_COLS + FBQ_FLOATS + FBD_FLOATS;
    for (int bank = 0; bank < fb_banks(M); ++bank) {
        const int m0 = bank == 0 ? 0 : fb_bank_lo(M), Mb = bank == 0 ? fb_bank_lo(M) : M - fb_bank_lo(M);
        hipLaunchKernelGGL(fb_from_points_kernel, dim3((work + 255) / 256 + 1), dim3(256), 0, stream, *pts, Mb, nyquist,
                           fbp + (size_t)bank * HOWL_FB_PACKED_FLOATS, m0, M == WIDE_MELS ? 1 + bank : 0);
    }
    HOWL_CHECK_LAUNCH("howl_fb_from_points");
    return HOWL_OK;
}

int howl_logmel_fwd(const float* pcm, int B, int L, long ld, const float* fbp, int M, float log_eps, const float* zmuv,
                    float* out, int layout, hipStream_t stream) {
    LogmelLaunch ll;
    const int rc = logmel_prepare(pcm, B, L, ld, fbp, M, log_eps, zmuv, out, layout, &ll);
    if (rc != HOWL_OK) return rc;
    // one persistent workgroup per CU (its LDS holds the tables once); each takes a contiguous share of the quads
    int grid = howl_num_cus();
    if (grid > ll.n_quads) grid = ll.n_quads;
    int waves = FE_WAVES;
    if (const char* e = getenv("HOWL_LOGMEL_WAVES")) waves = atoi(e) == 12 ? 12 : 16;   // occupancy experiments
    if (M == WIDE_MELS && getenv("HOWL_LOGMEL_TWO_LAUNCHES") == nullptr) {
        // the stock 80 mel bins: ONE pass over the spectrum, both banks' banded contractions on the powers in registers (round 6)
        HowlProfScope prof("logmel", stream);
        hipLaunchKernelGGL((logmel_kernel<12, NG_BANDED, true>), dim3((unsigned)grid), dim3(12 * 64), 0, stream, ll.pcm, ll.L, ll.ld, ll.T,
                           ll.total, fbp, M, ll.log_eps, ll.zmuv, out, ll.layout, ll.n_quads, ll.aligned, M);
        HOWL_CHECK_LAUNCH("howl_logmel_fwd");
        return HOWL_OK;
    }
    for (int bank = 0; bank < fb_banks(M); ++bank) {
        // more than 48 mel bins: one pass over the spectrum per bank, each writing its own columns of `out`
        const int m0 = bank == 0 ? 0 : fb_bank_lo(M);
        ll.fbp = fbp + (size_t)bank * HOWL_FB_PACKED_FLOATS;
        ll.M = bank == 0 ? fb_bank_lo(M) : M - fb_bank_lo(M);
        ll.out = out + (layout == 1 ? (long)m0 : (long)m0 * ll.T);
        HowlProfScope prof("logmel", stream);
        if (ll.M > 4 * NG_BANDED)
            hipLaunchKernelGGL((logmel_kernel<12, NG_MAX>), dim3((unsigned)grid), dim3(12 * 64), 0, stream, ll.pcm, ll.L, ll.ld, ll.T,
                               ll.total, ll.fbp, ll.M, ll.log_eps, ll.zmuv, ll.out, ll.layout, ll.n_quads, ll.aligned, ll.Mo);
        else if (waves == 16)
            hipLaunchKernelGGL((logmel_kernel<16, NG_BANDED>), dim3((unsigned)grid), dim3(16 * 64), 0, stream, ll.pcm, ll.L, ll.ld, ll.T,
                               ll.total, ll.fbp, ll.M, ll.log_eps, ll.zmuv, ll.out, ll.layout, ll.n_quads, ll.aligned, ll.Mo);
        else
            hipLaunchKernelGGL((logmel_kernel<12, NG_BANDED>), dim3((unsigned)grid), dim3(12 * 64), 0, stream, ll.pcm, ll.L, ll.ld, ll.T,
                               ll.total, ll.fbp, ll.M, ll.log_eps, ll.zmuv, ll.out, ll.layout, ll.n_quads, ll.aligned, ll.Mo);
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
