// MobileNetClassifier ("mobilenet"; reference howl/model/cnn.py:15-29, BASELINE configs[4]): forward and backward.
//
//   downsample = Conv2d(1,3,3,padding=(1,3)) + BatchNorm2d(3) + ReLU + MaxPool2d((1,2))          (cnn.py:18-21)
//   model      = torchvision mobilenet_v2 with a num_labels classifier                           (cnn.py:22-24)
//   forward(x) = model(downsample(x[:, :1]))                                                     (cnn.py:26-29)
// torchvision is a third-party dependency that is not under /root/reference; the body follows the published
// architecture (Sandler et al. 2018 table 2 == torchvision mobilenetv2.py, width 1.0): see oracle/mobilenet.py.
//
// Design.  Activations are channels-last, one row per pixel: tensor k is an (M_k = B*H_k*W_k) x C_k row-major matrix.
//   * 1x1 convolutions (35 of the 53) ARE matrix products on that layout -> the fp32 MFMA GEMM (howl_gemm.hip.h),
//     forward, data gradient and weight gradient (split-K + fixed-order slab sum); the two dense 3x3 convolutions
//     (1->3 and 3->32) go through an im2col matrix and the same GEMM;
//   * depthwise 3x3, BatchNorm statistics / apply / backward, ReLU6, pooling are bandwidth-bound row sweeps with the
//     channel as the unit-stride index (coalesced at any C); per-channel reductions are two-stage and fixed-order
//     (fp64 partials), so results do not depend on scheduling;
//   * the layer table is built once on the host and published through howl_mobilenet_layer(): the Python module lays
//     its parameters out in ONE flat buffer at those offsets, in PyTorch's own shapes, so gradients land in a flat
//     buffer of the same layout and the optimiser is a single fused AdamW launch.
// Saved for the backward pass (in the caller's workspace): each layer's convolution output z_k, its batch statistics
// and its output y_k; ReLU6 masks and normalised values are recomputed from z_k.
#include <algorithm>
#include <vector>

#include "howl_common.hip.h"
#include "../../include/howl_hip.h"
#include "howl_gemm.hip.h"

namespace {

enum { MB_DENSE3 = 0, MB_PW = 1, MB_DW = 2 };
enum { MB_ACT_NONE = 0, MB_ACT_RELU6 = 1, MB_ACT_RELU = 2 };
constexpr int MB_LAST = 1280;
constexpr float MB_EPS = 1e-5f;
constexpr double MB_MOMENTUM = 0.1;
constexpr int MB_CHUNKS = 512;      // upper bound on the row chunks of the two-stage column reductions
constexpr int MB_WGRAD_SPLITS = 1024;  // split-K bound of the weight-gradient GEMMs (long, thin reductions over pixels)

struct Net {
    std::vector<HowlMbLayer> layers;
    size_t feature_params = 0;  // floats before the classifier
    size_t buffers = 0;         // floats of BN running statistics
};

Net build_net() {
    Net n;
    size_t po = 0, bo = 0;
    auto add = [&](int kind, int cin, int cout, int stride, int ph, int pw, int act, int bias, int pool, int res_src, int feat,
                   int sub, int wrapped) {
        HowlMbLayer l{};
        l.kind = kind;
        l.cin = cin;
        l.cout = cout;
        l.stride = stride;
        l.pad_h = ph;
        l.pad_w = pw;
        l.act = act;
        l.bias = bias;
        l.pool = pool;
        l.res_src = res_src;
        l.feat = feat;
        l.sub = sub;
        l.wrapped = wrapped;
        const size_t wn = (kind == MB_DENSE3) ? (size_t)cout * cin * 9 : (kind == MB_DW ? (size_t)cout * 9 : (size_t)cout * cin);
        l.w_off = (long long)po;
        po += wn;
        l.b_off = -1;
        if (bias) {
            l.b_off = (long long)po;
            po += cout;
        }
        l.gamma_off = (long long)po;
        po += cout;
        l.beta_off = (long long)po;
        po += cout;
        l.rmean_off = (long long)bo;
        bo += cout;
        l.rvar_off = (long long)bo;
        bo += cout;
        n.layers.push_back(l);
    };
    add(MB_DENSE3, 1, 3, 1, 1, 3, MB_ACT_RELU, 1, 1, -1, -1, -1, 0);     // downsample (cnn.py:18-21)
    add(MB_DENSE3, 3, 32, 2, 1, 1, MB_ACT_RELU6, 0, 0, -1, 0, -1, 1);    // features[0]
    static const int setting[7][4] = {{1, 16, 1, 1}, {6, 24, 2, 2}, {6, 32, 3, 2}, {6, 64, 4, 2},
                                      {6, 96, 3, 1}, {1 * 6, 160, 3, 2}, {6, 320, 1, 1}};
    int inp = 32, feat = 1;
    for (const auto& st : setting) {
        const int t = st[0], c = st[1], reps = st[2], s = st[3];
        for (int i = 0; i < reps; ++i) {
            const int stride = (i == 0) ? s : 1;
            const int hidden = inp * t;
            const int block_in = (int)n.layers.size() - 1;  // the layer whose output enters this block
            int j = 0;
            if (t != 1) {
                add(MB_PW, inp, hidden, 1, 0, 0, MB_ACT_RELU6, 0, 0, -1, feat, 0, 1);
                j = 1;
            }
            add(MB_DW, hidden, hidden, stride, 1, 1, MB_ACT_RELU6, 0, 0, -1, feat, j, 1);
            add(MB_PW, hidden, c, 1, 0, 0, MB_ACT_NONE, 0, 0, (stride == 1 && inp == c) ? block_in : -1, feat, j + 1, 0);
            inp = c;
            ++feat;
        }
    }
    add(MB_PW, inp, MB_LAST, 1, 0, 0, MB_ACT_RELU6, 0, 0, -1, 18, -1, 1);  // features[18]
    n.feature_params = po;
    n.buffers = bo;
    return n;
}

const Net& net() {
    static const Net n = build_net();
    return n;
}

// ---------------------------------------------------------------------------------------------------------
// geometry and workspace plan for one (B, H0, W0)
// ---------------------------------------------------------------------------------------------------------
struct Geo {
    int hin, win, ho, wo;  // convolution input / output extent
    int hy, wy;            // layer output extent (after the optional (1,2) max pool)
    long mz, my;           // rows of z_k and y_k
};

struct Plan {
    std::vector<Geo> g;
    std::vector<size_t> z, y, yp, stats, dact;  // float offsets
    size_t dz = 0, dz2 = 0, col = 0, dcol = 0, part = 0, gemm_scratch = 0, m12 = 0, pooled = 0, pooled_d = 0, dpooled = 0, dyp = 0;
    size_t total_floats = 0;
};

int wgrad_splits(long rows) {  // must match wgrad_gemm() in howl_gemm.hip.h
    long s = rows / 512;
    return (int)(s < 1 ? 1 : (s > MB_WGRAD_SPLITS ? MB_WGRAD_SPLITS : s));
}

// Column reductions: a wave covers 64 / Cp rows at a time when the matrix is narrow (Cp = C rounded up to a power of
// two <= 32), else 64 columns of one row; the rows are cut into chunks so that the launch fills the chip.
inline int col_pack(int C) {
    if (C > 32) return 64;
    int cp = 1;
    while (cp < C) cp <<= 1;
    return cp;
}
inline int chunks_for(long rows, int C) {
    const long rows_per_wave_iter = 64 / col_pack(C);
    long c = rows / (64 * rows_per_wave_iter);   // >= 16 row-iterations per wave
    return (int)(c < 1 ? 1 : (c > MB_CHUNKS ? MB_CHUNKS : c));
}

Plan make_plan(int B, int H0, int W0, int num_labels) {
    const Net& n = net();
    Plan p;
    size_t off = 0;
    auto take = [&](size_t floats) {
        const size_t o = off;
        off += (floats + 63) / 64 * 64;  // 256-byte granules
        return o;
    };
    int h = H0, w = W0;
    size_t max_dz = 0, max_col = 0, max_scr = (size_t)64 * num_labels * MB_LAST, max_part = 0;
    for (const HowlMbLayer& l : n.layers) {
        Geo g{};
        g.hin = h;
        g.win = w;
        if (l.kind == MB_PW) {
            g.ho = h;
            g.wo = w;
        } else {
            g.ho = (h + 2 * l.pad_h - 3) / l.stride + 1;
            g.wo = (w + 2 * l.pad_w - 3) / l.stride + 1;
        }
        g.hy = g.ho;
        g.wy = l.pool ? g.wo / 2 : g.wo;
        g.mz = (long)B * g.ho * g.wo;
        g.my = (long)B * g.hy * g.wy;
        p.g.push_back(g);
        p.z.push_back(take((size_t)g.mz * l.cout));
        p.yp.push_back(l.pool ? take((size_t)g.mz * l.cout) : 0);
        p.y.push_back(take((size_t)g.my * l.cout));
        p.stats.push_back(take(2 * (size_t)l.cout));
        p.dact.push_back(take((size_t)g.my * l.cout));
        max_dz = std::max(max_dz, (size_t)g.mz * l.cout);
        if (l.kind == MB_DENSE3) max_col = std::max(max_col, (size_t)g.mz * 9 * l.cin);
        const size_t wn = (l.kind == MB_DENSE3) ? (size_t)l.cout * l.cin * 9 : (size_t)l.cout * l.cin;
        if (l.kind != MB_DW) max_scr = std::max(max_scr, (size_t)wgrad_splits(g.mz) * wn);
        if (l.kind == MB_DW) max_scr = std::max(max_scr, (size_t)chunks_for(g.mz, 64) * l.cout * 9);
        max_part = std::max(max_part, (size_t)chunks_for(g.mz, l.cout) * 2 * l.cout * 2);
        h = g.hy;
        w = g.wy;
    }
    p.dz = take(max_dz);
    p.dz2 = take(max_dz);
    p.dyp = take(max_dz);
    p.col = take(max_col);
    p.dcol = take(max_col);
    p.part = take(max_part);  // doubles
    p.gemm_scratch = take(max_scr);
    p.m12 = take(2 * MB_LAST);
    p.pooled = take((size_t)B * MB_LAST);
    p.pooled_d = take((size_t)B * MB_LAST);
    p.dpooled = take((size_t)B * MB_LAST);
    p.total_floats = off;
    return p;
}

// ---------------------------------------------------------------------------------------------------------
// kernels (channels-last; `c` is always the unit-stride index)
// ---------------------------------------------------------------------------------------------------------
// channel of flat element idx of an (M x C) matrix: 32-bit arithmetic whenever the index fits (a 64-bit % is ~100
// instructions, more than the rest of an elementwise kernel)
__device__ __forceinline__ int chan_of(long idx, int C) {
    return idx < (1L << 31) ? (int)((unsigned)idx % (unsigned)C) : (int)(idx % C);
}

// col[m][c*9 + tap] = x[b, oh*s - ph + tap/3, ow*s - pw + tap%3, c]  (zero outside); x addressed through strides so
// that the network input can be a (B,1,M,T) view of a multi-channel feature tensor
__global__ void im2col3x3_kernel(const float* __restrict__ x, long sb, long sh, long sw, long sc, int H, int W, int C, int Ho,
                                 int Wo, int stride, int ph, int pw, long total, float* __restrict__ col) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int K = 9 * C;
        const long m = idx / K;
        const int k = (int)(idx - m * K);
        const int c = k / 9, tap = k - 9 * c;
        const long b = m / ((long)Ho * Wo);
        const int r = (int)(m - b * Ho * Wo);
        const int oh = r / Wo, ow = r - oh * Wo;
        const int ih = oh * stride - ph + tap / 3, iw = ow * stride - pw + tap % 3;
        float v = 0.0f;
        if (ih >= 0 && ih < H && iw >= 0 && iw < W) v = x[b * sb + ih * sh + iw * sw + c * sc];
        col[idx] = v;
    }
}

// dx[b,ih,iw,c] = sum over the (oh,ow,tap) that read it of dcol[m][c*9 + tap]   (gather form: no atomics)
__global__ void col2im3x3_kernel(const float* __restrict__ dcol, int H, int W, int C, int Ho, int Wo, int stride, int ph, int pw,
                                 long total, float* __restrict__ dx) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = chan_of(idx, C);
        const long pix = idx / C;
        const int iw = (int)(pix % W);
        const long t = pix / W;
        const int ih = (int)(t % H);
        const long b = t / H;
        float acc = 0.0f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int nh = ih + ph - kh;
            if (nh < 0 || nh % stride != 0) continue;
            const int oh = nh / stride;
            if (oh >= Ho) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int nw = iw + pw - kw;
                if (nw < 0 || nw % stride != 0) continue;
                const int ow = nw / stride;
                if (ow >= Wo) continue;
                acc += dcol[((b * Ho + oh) * Wo + ow) * (9L * C) + c * 9 + kh * 3 + kw];
            }
        }
        dx[idx] = acc;
    }
}

// depthwise 3x3, padding 1: z[b,oh,ow,c] = sum_tap w[c*9+tap] * x[b, oh*s-1+kh, ow*s-1+kw, c]
// A thread produces DW_SEG = 4 consecutive outputs along W for one channel from ONE window of inputs (3 rows x 6 columns at
// stride 1, 3 x 9 at stride 2): 4.5 / 6.75 loads per output instead of 9 (+ the nine weights once per thread instead of once
// per output); a one-output-per-thread version was bound by the number of vector-memory instructions, not by bytes.  All
// loads are unconditional from clamped addresses and zeroed afterwards (a load inside an `if` is followed by its own wait).
// Every output accumulates its nine taps in the order (kh, kw) with fmaf, exactly like the scalar version.
// Launch geometry: blockIdx.x = image row (b, oh), blockIdx.y * 256 + threadIdx.x = (segment of 4 columns, c), channel
// fastest: a wave's loads are 256 contiguous bytes per tap.
constexpr int DW_SEG = 4;
template <int STRIDE>
__global__ __launch_bounds__(256) void dw3x3_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, int H, int W,
                                                        int C, int Ho, int Wo, float* __restrict__ z) {
    constexpr int NIN = (DW_SEG - 1) * STRIDE + 3;     // input columns under DW_SEG outputs
    const int nseg = (Wo + DW_SEG - 1) / DW_SEG;
    const int rc = blockIdx.y * 256 + threadIdx.x;
    if (rc >= nseg * C) return;
    const int seg = rc / C, c = rc - seg * C;
    const int ow0 = seg * DW_SEG;
    const int b = blockIdx.x / Ho, oh = blockIdx.x - b * Ho;
    const float* xb = x + (long)b * H * W * C + c;
    float wk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k] = w[c * 9 + k];
    float v[3][NIN];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int ih = oh * STRIDE - 1 + kh;
        const int ihc = min(max(ih, 0), H - 1);
#pragma unroll
        for (int j = 0; j < NIN; ++j) {
            const int iw = ow0 * STRIDE - 1 + j;
            const int iwc = min(max(iw, 0), W - 1);
            v[kh][j] = xb[((long)ihc * W + iwc) * C];
        }
#pragma unroll
        for (int j = 0; j < NIN; ++j) {
            const int iw = ow0 * STRIDE - 1 + j;
            if (ih < 0 || ih >= H || iw < 0 || iw >= W) v[kh][j] = 0.0f;
        }
    }
    float* zr = z + (long)blockIdx.x * Wo * C + c;
#pragma unroll
    for (int o = 0; o < DW_SEG; ++o) {
        float acc = 0.0f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) acc = fmaf(wk[kh * 3 + kw], v[kh][o * STRIDE + kw], acc);
        if (ow0 + o < Wo) zr[(long)(ow0 + o) * C] = acc;
    }
}

// dx[b,ih,iw,c] = sum_tap w[c*9+tap] * dz[b, oh, ow, c] over the outputs (oh, ow) whose window holds (ih, iw) at that tap:
// oh*s - 1 + kh = ih, ow*s - 1 + kw = iw.  Same thread geometry over the INPUT positions (4 consecutive iw per thread); the
// gradient window under them is 3 rows x 6 columns at stride 1 and 2 x 3 at stride 2 (loaded for every tap parity, masked).
template <int STRIDE>
__global__ __launch_bounds__(256) void dw3x3_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ w, int H,
                                                          int W, int C, int Ho, int Wo, float* __restrict__ dx) {
    const int nseg = (W + DW_SEG - 1) / DW_SEG;
    const int rc = blockIdx.y * 256 + threadIdx.x;
    if (rc >= nseg * C) return;
    const int seg = rc / C, c = rc - seg * C;
    const int iw0 = seg * DW_SEG;
    const int b = blockIdx.x / H, ih = blockIdx.x - b * H;
    const float* zb = dz + (long)b * Ho * Wo * C + c;
    float wk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k] = w[c * 9 + k];
    // gradient columns that can touch inputs iw0 .. iw0+3: ow = (iw + 1 - kw) / s  ->  from (iw0 - 1) / s (floor) upwards
    constexpr int NCOL = STRIDE == 1 ? DW_SEG + 2 : (DW_SEG + 1) / STRIDE + 2;   // 6 at stride 1, 4 at stride 2
    const int owb = (iw0 - 1 + STRIDE) / STRIDE - 1;               // floor((iw0 - 1) / s) for iw0 >= 0
    float v[3][NCOL];
    bool okh[3];
    int ohs[3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int nh = ih + 1 - kh;                                // = oh * s
        const int oh = (nh + STRIDE) / STRIDE - 1;                 // floor division for nh >= -s
        okh[kh] = nh >= 0 && oh * STRIDE == nh && oh < Ho;
        ohs[kh] = min(max(oh, 0), Ho - 1);
#pragma unroll
        for (int j = 0; j < NCOL; ++j) {
            const int owc = min(max(owb + j, 0), Wo - 1);
            v[kh][j] = zb[((long)ohs[kh] * Wo + owc) * C];
        }
    }
    float* xr = dx + (long)blockIdx.x * W * C + c;
#pragma unroll
    for (int o = 0; o < DW_SEG; ++o) {
        const int iw = iw0 + o;
        float acc = 0.0f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int nw = iw + 1 - kw;                        // = ow * s
                const int ow = (nw + STRIDE) / STRIDE - 1;
                const bool ok = okh[kh] && nw >= 0 && ow * STRIDE == nw && ow < Wo;
                // window slot of ow; clamped only to keep the (masked) register index inside the array
                const int j = min(max(ow - owb, 0), NCOL - 1);
                float g = v[kh][0];
#pragma unroll
                for (int q = 1; q < NCOL; ++q) g = (j == q) ? v[kh][q] : g;
                acc = fmaf(ok ? wk[kh * 3 + kw] : 0.0f, g, acc);
            }
        if (iw < W) xr[(long)iw * C] = acc;
    }
}

// dW[c][tap] = sum_{b,oh,ow} dz[.,c] * x[shifted, c]: block = 64 channels x one chunk of IMAGE rows (b, oh); its 4 waves
// split the rows, and a lane (= channel) walks a row in segments of DW_SEG = 4 outputs that share one input window
// (3 rows x 6 columns at stride 1, 3 x 9 at stride 2: 5.5 / 7.75 loads per pixel instead of 10, all unconditional from
// clamped addresses).  part[chunk][c*9+tap]; the chunks are folded in a fixed order by sum_slabs_kernel.
template <int STRIDE>
__global__ __launch_bounds__(256) void dw3x3_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ x, int H,
                                                          int W, int C, int Ho, int Wo, int nrows, int rows_per_chunk,
                                                          float* __restrict__ part) {
    constexpr int NIN = (DW_SEG - 1) * STRIDE + 3;
    __shared__ float red[4][9][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * rows_per_chunk;
    const int r1 = nrows < r0 + rows_per_chunk ? nrows : r0 + rows_per_chunk;
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.0f;
    if (c < C) {
        for (int t = r0 + rg; t < r1; t += 4) {
            const int b = t / Ho, oh = t - b * Ho;
            const float* xb = x + (long)b * H * W * C + c;
            const float* zr = dz + (long)t * Wo * C + c;
            for (int ow0 = 0; ow0 < Wo; ow0 += DW_SEG) {
                float g[DW_SEG], v[3][NIN];
#pragma unroll
                for (int o = 0; o < DW_SEG; ++o) g[o] = zr[(long)min(ow0 + o, Wo - 1) * C];
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int ihc = min(max(oh * STRIDE - 1 + kh, 0), H - 1);
#pragma unroll
                    for (int j = 0; j < NIN; ++j) {
                        const int iwc = min(max(ow0 * STRIDE - 1 + j, 0), W - 1);
                        v[kh][j] = xb[((long)ihc * W + iwc) * C];
                    }
                }
#pragma unroll
                for (int o = 0; o < DW_SEG; ++o)
                    if (ow0 + o >= Wo) g[o] = 0.0f;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int ih = oh * STRIDE - 1 + kh;
#pragma unroll
                    for (int j = 0; j < NIN; ++j) {
                        const int iw = ow0 * STRIDE - 1 + j;
                        if (ih < 0 || ih >= H || iw < 0 || iw >= W) v[kh][j] = 0.0f;
                    }
                }
#pragma unroll
                for (int o = 0; o < DW_SEG; ++o)
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) acc[kh * 3 + kw] = fmaf(g[o], v[kh][o * STRIDE + kw], acc[kh * 3 + kw]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) red[rg][t][lane] = acc[t];
    __syncthreads();
    if (rg == 0 && c < C) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
            part[(size_t)blockIdx.y * C * 9 + c * 9 + t] = ((red[0][t][lane] + red[1][t][lane]) + red[2][t][lane]) + red[3][t][lane];
    }
}

__device__ __forceinline__ float mb_act(float v, int act) {
    if (act == MB_ACT_RELU6) return fminf(fmaxf(v, 0.0f), 6.0f);
    if (act == MB_ACT_RELU) return fmaxf(v, 0.0f);
    return v;
}
__device__ __forceinline__ bool mb_act_passes(float v, int act) {  // derivative of the activation is 1
    if (act == MB_ACT_RELU6) return v > 0.0f && v < 6.0f;
    if (act == MB_ACT_RELU) return v > 0.0f;
    return true;
}

// Two-stage per-channel reductions over the rows of an (M x C) matrix.  MODE 0: (sum z, sum z^2).
// MODE 1 (BatchNorm backward): g = dy * act'(gamma*xhat + beta), xhat = (z - mean) * rstd -> (sum g, sum g*xhat).
// Lane -> (row slot rs = lane / cp, column blockIdx.x*64 + lane % cp): cp = 64 for wide matrices, a power of two <= 32
// for narrow ones so that no lane idles on the 3..32-channel layers that have the most rows.  Rows advance by
// 4 waves x (64/cp) slots; four independent accumulator pairs keep four loads in flight.
template <int MODE>
__global__ __launch_bounds__(256) void col_reduce_kernel(const float* __restrict__ z, const float* __restrict__ dy,
                                                         const float* __restrict__ stats, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int act, long rows, int C, int cp,
                                                         long rows_per_chunk, double* __restrict__ part) {
    __shared__ double red[2][4][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int rsub = 64 / cp;
    const int cl = lane & (cp - 1), rs = lane / cp;
    const int c = blockIdx.x * 64 + cl;
    const long r0 = (long)blockIdx.y * rows_per_chunk;
    const long r1 = rows < r0 + rows_per_chunk ? rows : r0 + rows_per_chunk;
    double s0 = 0.0, s1 = 0.0;
    if (c < C && cl < cp) {
        float mean = 0.0f, rstd = 1.0f, ga = 1.0f, be = 0.0f;
        if (MODE == 1) {
            mean = stats[c];
            rstd = stats[C + c];
            ga = gamma[c];
            be = beta[c];
        }
        const long step = 4L * rsub;
        float a0[4], a1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a0[u] = a1[u] = 0.0f;
        long m = r0 + (long)rg * rsub + rs;
        int since_flush = 0;
        for (; m < r1; m += 4 * step) {
            float v[4], d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long mm = m + u * step;
                const bool ok = mm < r1;
                const long idx = (ok ? mm : m) * C + c;
                v[u] = z[idx];
                d[u] = MODE == 1 ? dy[idx] : 0.0f;
                if (!ok) {
                    v[u] = MODE == 1 ? mean : 0.0f;   // contributes nothing
                    d[u] = 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (MODE == 0) {
                    a0[u] += v[u];
                    a1[u] = fmaf(v[u], v[u], a1[u]);
                } else {
                    const float xh = (v[u] - mean) * rstd;
                    const float g = mb_act_passes(fmaf(ga, xh, be), act) ? d[u] : 0.0f;
                    a0[u] += g;
                    a1[u] = fmaf(g, xh, a1[u]);
                }
            }
            if (++since_flush == 16) {   // short fp32 runs, fp64 totals
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    s0 += (double)a0[u];
                    s1 += (double)a1[u];
                    a0[u] = a1[u] = 0.0f;
                }
                since_flush = 0;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s0 += (double)a0[u];
            s1 += (double)a1[u];
        }
    }
    red[0][rg][lane] = s0;
    red[1][rg][lane] = s1;
    __syncthreads();
    if (rg == 0 && rs == 0 && c < C) {
        double t0 = 0.0, t1 = 0.0;
        for (int w = 0; w < 4; ++w)
            for (int q = 0; q < rsub; ++q) {
                t0 += red[0][w][q * cp + cl];
                t1 += red[1][w][q * cp + cl];
            }
        part[((size_t)blockIdx.y * 2 + 0) * C + c] = t0;
        part[((size_t)blockIdx.y * 2 + 1) * C + c] = t1;
    }
}

// folds the chunk partials of 64 columns: the 4 waves split the chunks, lanes are columns (coalesced), fixed order
constexpr int FIN_RG = 16;   // row groups of a finalize block (1024 threads): <= 512 chunk rows are two trips of 16 loads
constexpr int FIN_THREADS = 64 * FIN_RG;
__device__ __forceinline__ void fold_chunks(const double* __restrict__ part, int chunks, int C, int c, int rg, double& s0,
                                            double& s1, double (&red)[2][FIN_RG][64], int lane) {
    // Sixteen row groups x eight rows per trip, the 16 loads of a trip in flight together (clamped row, masked sum).  With 4
    // row groups this was 8+ dependent trips to L2 in a kernel that is nothing but latency (53 + 53 launches per step).
    double a0 = 0.0, a1 = 0.0;
    const int cc = c < C ? c : C - 1;
    for (int k0 = rg; k0 < chunks; k0 += 8 * FIN_RG) {
        double v0[8], v1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + u * FIN_RG < chunks ? k0 + u * FIN_RG : chunks - 1;
            v0[u] = part[((size_t)k * 2 + 0) * C + cc];
            v1[u] = part[((size_t)k * 2 + 1) * C + cc];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool ok = k0 + u * FIN_RG < chunks;
            a0 += ok ? v0[u] : 0.0;
            a1 += ok ? v1[u] : 0.0;
        }
    }
    red[0][rg][lane] = a0;
    red[1][rg][lane] = a1;
    __syncthreads();
    s0 = 0.0;
    s1 = 0.0;
    if (rg == 0) {
#pragma unroll
        for (int g = 0; g < FIN_RG; ++g) {
            s0 += red[0][g][lane];
            s1 += red[1][g][lane];
        }
    }
}

// batch statistics (biased variance for normalisation, unbiased for the running estimate: nn.BatchNorm2d)
__global__ __launch_bounds__(FIN_THREADS) void bn_stats_finalize_kernel(const double* __restrict__ part, int chunks, int C,
                                                                double count, float* __restrict__ stats,
                                                                float* __restrict__ rmean, float* __restrict__ rvar) {
    __shared__ double red[2][FIN_RG][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    double s0, s1;
    fold_chunks(part, chunks, C, c, rg, s0, s1, red, lane);
    if (rg != 0 || c >= C) return;
    const double mean = s0 / count;
    double var = s1 / count - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    stats[c] = (float)mean;
    stats[C + c] = (float)(1.0 / sqrt(var + (double)MB_EPS));
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    rmean[c] = (float)((1.0 - MB_MOMENTUM) * (double)rmean[c] + MB_MOMENTUM * mean);
    rvar[c] = (float)((1.0 - MB_MOMENTUM) * (double)rvar[c] + MB_MOMENTUM * unbiased);
}

__global__ void bn_eval_stats_mb_kernel(const float* __restrict__ rmean, const float* __restrict__ rvar, int C,
                                        float* __restrict__ stats) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    stats[c] = rmean[c];
    stats[C + c] = 1.0f / sqrtf(rvar[c] + MB_EPS);
}

// y = act(gamma * (z - mean) * rstd + beta) [+ res]
__global__ void bn_act_fwd_kernel(const float* __restrict__ z, const float* __restrict__ stats, const float* __restrict__ gamma,
                                  const float* __restrict__ beta, const float* __restrict__ res, int act, int C, long total,
                                  float* __restrict__ y) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = chan_of(idx, C);
        float v = mb_act(fmaf(gamma[c], (z[idx] - stats[c]) * stats[C + c], beta[c]), act);
        if (res != nullptr) v += res[idx];
        y[idx] = v;
    }
}

// dgamma = sum g*xhat, dbeta = sum g; m1 = dbeta / M, m2 = dgamma / M for the apply pass
__global__ __launch_bounds__(FIN_THREADS) void bn_bwd_finalize_mb_kernel(const double* __restrict__ part, int chunks, int C,
                                                                 double count, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, float* __restrict__ m12) {
    __shared__ double red[2][FIN_RG][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    double s0, s1;
    fold_chunks(part, chunks, C, c, rg, s0, s1, red, lane);
    if (rg != 0 || c >= C) return;
    dbeta[c] = (float)s0;
    dgamma[c] = (float)s1;
    m12[c] = (float)(s0 / count);
    m12[C + c] = (float)(s1 / count);
}

// column sums only (conv-bias gradient): out[c] = sum over chunks of part[k][0][c]
__global__ __launch_bounds__(FIN_THREADS) void colsum_finalize_kernel(const double* __restrict__ part, int chunks, int C,
                                                              float* __restrict__ out) {
    __shared__ double red[2][FIN_RG][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    double s0, s1;
    fold_chunks(part, chunks, C, c, rg, s0, s1, red, lane);
    if (rg == 0 && c < C) out[c] = (float)s0;
}

// dz = gamma * rstd * (g - m1 - xhat * m2)
__global__ void bn_bwd_apply_kernel(const float* __restrict__ z, const float* __restrict__ dy, const float* __restrict__ stats,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ m12, int act, int C, long total, float* __restrict__ dz) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = chan_of(idx, C);
        const float rstd = stats[C + c];
        const float xh = (z[idx] - stats[c]) * rstd;
        const float g = mb_act_passes(fmaf(gamma[c], xh, beta[c]), act) ? dy[idx] : 0.0f;
        dz[idx] = gamma[c] * rstd * (g - m12[c] - xh * m12[C + c]);
    }
}

// MaxPool2d((1,2)) over W (floor), channels-last
__global__ void maxpool12_fwd_kernel(const float* __restrict__ x, int W, int Wp, int C, long total, float* __restrict__ y) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = chan_of(idx, C);
        const long pix = idx / C;
        const int wp = (int)(pix % Wp);
        const long bh = pix / Wp;
        const float a = x[(bh * W + 2 * wp) * C + c], b = x[(bh * W + 2 * wp + 1) * C + c];
        y[idx] = fmaxf(a, b);
    }
}
// the gradient goes to the first maximal element (PyTorch's argmax rule); columns past 2*Wp get none
__global__ void maxpool12_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int W, int Wp, int C, long total,
                                     float* __restrict__ dx) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = chan_of(idx, C);
        const long pix = idx / C;
        const int w = (int)(pix % W);
        const long bh = pix / W;
        const int wp = w >> 1;
        float g = 0.0f;
        if (wp < Wp) {
            const float a = x[(bh * W + 2 * wp) * C + c], b = x[(bh * W + 2 * wp + 1) * C + c];
            const bool first = a >= b;  // ties: the earlier index wins
            if ((w & 1) == 0 ? first : !first) g = dy[(bh * Wp + wp) * C + c];
        }
        dx[idx] = g;
    }
}

// adaptive_avg_pool2d(1): pooled[b][c] = mean over the HW pixels; optional dropout mask applied to a second output
__global__ void avgpool_fwd_kernel(const float* __restrict__ x, int HW, int C, const float* __restrict__ mask, float scale,
                                   long total, float* __restrict__ pooled, float* __restrict__ pooled_d) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = chan_of(idx, C);
        const long b = idx / C;
        float acc = 0.0f;
        for (int p = 0; p < HW; ++p) acc += x[(b * HW + p) * C + c];
        const float v = acc / (float)HW;
        pooled[idx] = v;
        pooled_d[idx] = mask != nullptr ? v * mask[idx] * scale : v;
    }
}
__global__ void avgpool_bwd_kernel(const float* __restrict__ dpooled, int HW, int C, const float* __restrict__ mask, float scale,
                                   long total, float* __restrict__ dx) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = chan_of(idx, C);
        const long b = idx / ((long)HW * C);
        float g = dpooled[b * C + c];
        if (mask != nullptr) g *= mask[b * C + c] * scale;
        dx[idx] = g / (float)HW;
    }
}

__global__ void add_inplace_kernel(float* __restrict__ a, const float* __restrict__ b, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) a[i] += b[i];
}

inline unsigned flat_grid(long total) {
    long blocks = (total + 255) / 256;
    return (unsigned)(blocks < 1 ? 1 : (blocks > 65536 ? 65536 : blocks));
}

struct Ctx {
    const Net* n;
    Plan p;
    float* ws;
    const float* params;
    hipStream_t s;
    int B;
};

// convolution of layer k: z = conv(in)
void conv_forward(const Ctx& c, int k, const float* in, long sb, long sh, long sw, long sc, float* z) {
    const HowlMbLayer& l = c.n->layers[k];
    const Geo& g = c.p.g[k];
    const float* w = c.params + l.w_off;
    if (l.kind == MB_PW) {
        gemm(c.s, true, in, lin(l.cin), 1, lin(0), w, lin(1), l.cin, (int)g.mz, l.cout, l.cin, 1, nullptr, 0, z, l.cout, 0);
    } else if (l.kind == MB_DW) {
        const dim3 grid((unsigned)(c.B * g.ho), (((g.wo + DW_SEG - 1) / DW_SEG) * l.cin + 255) / 256);
        if (l.stride == 1)
            hipLaunchKernelGGL(dw3x3_fwd_kernel<1>, grid, dim3(256), 0, c.s, in, w, g.hin, g.win, l.cin, g.ho, g.wo, z);
        else
            hipLaunchKernelGGL(dw3x3_fwd_kernel<2>, grid, dim3(256), 0, c.s, in, w, g.hin, g.win, l.cin, g.ho, g.wo, z);
    } else {
        float* col = c.ws + c.p.col;
        const int K = 9 * l.cin;
        const long total = g.mz * K;
        hipLaunchKernelGGL(im2col3x3_kernel, dim3(flat_grid(total)), dim3(256), 0, c.s, in, sb, sh, sw, sc, g.hin, g.win, l.cin,
                           g.ho, g.wo, l.stride, l.pad_h, l.pad_w, total, col);
        gemm(c.s, true, col, lin(K), 1, lin(0), w, lin(1), K, (int)g.mz, l.cout, K, 1, l.bias ? c.params + l.b_off : nullptr, 0, z,
             l.cout, 0);
    }
}

}  // namespace

extern "C" {

size_t howl_mobilenet_num_layers(void) { return net().layers.size(); }

int howl_mobilenet_layer(int i, HowlMbLayer* out) {
    HOWL_REQUIRE(out != nullptr, "howl_mobilenet_layer: null pointer");
    HOWL_REQUIRE(i >= 0 && i < (int)net().layers.size(), "howl_mobilenet_layer: index out of range");
    *out = net().layers[i];
    return HOWL_OK;
}

size_t howl_mobilenet_param_floats(int num_labels) {
    return net().feature_params + (size_t)num_labels * MB_LAST + (size_t)num_labels;
}

size_t howl_mobilenet_buffer_floats(void) { return net().buffers; }

size_t howl_mobilenet_workspace_bytes(int B, int M, int T, int num_labels) {
    if (B < 1 || M < 1 || T < 1 || num_labels < 1) return 0;
    return make_plan(B, M, T, num_labels).total_floats * sizeof(float) + 256;
}

int howl_mobilenet_fwd(const float* params, float* buffers, int num_labels, const float* x, long sb, long sm, long st, int B,
                       int M, int T, int training, const float* drop_mask, float drop_scale, float* logits, void* ws,
                       size_t ws_bytes, hipStream_t stream) {
    HOWL_REQUIRE(params && buffers && x && logits && ws, "howl_mobilenet_fwd: null pointer");
    HOWL_REQUIRE(B >= 1 && M >= 1 && T >= 1 && num_labels >= 1, "howl_mobilenet_fwd: bad shape");
    HOWL_REQUIRE(ws_bytes >= howl_mobilenet_workspace_bytes(B, M, T, num_labels), "howl_mobilenet_fwd: workspace too small");
    Ctx c{&net(), make_plan(B, M, T, num_labels), reinterpret_cast<float*>(ws), params, stream, B};
    const int nl = (int)c.n->layers.size();
    HOWL_REQUIRE(c.p.g[nl - 1].hy >= 1 && c.p.g[nl - 1].wy >= 1, "howl_mobilenet_fwd: input too small for the network");
    double* part = reinterpret_cast<double*>(c.ws + c.p.part);
    const float* in = x;
    long isb = sb, ish = sm, isw = st, isc = 0;  // network input: (B, 1, M, T) view, H = mel, W = time
    for (int k = 0; k < nl; ++k) {
        const HowlMbLayer& l = c.n->layers[k];
        const Geo& g = c.p.g[k];
        float* z = c.ws + c.p.z[k];
        float* stats = c.ws + c.p.stats[k];
        conv_forward(c, k, in, isb, ish, isw, isc, z);
        if (training) {
            const int chunks = chunks_for(g.mz, l.cout);
            const long rpc = (g.mz + chunks - 1) / chunks;
            {
                HowlProfScope prof("mb_sweep", stream, 4.0 * (double)g.mz * l.cout);     // reads z once
                hipLaunchKernelGGL(col_reduce_kernel<0>, dim3((l.cout + 63) / 64, chunks), dim3(256), 0, stream, (const float*)z,
                                   (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                                   0, g.mz, l.cout, col_pack(l.cout), rpc, part);
            }
            hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((l.cout + 63) / 64), dim3(FIN_THREADS), 0, stream, (const double*)part,
                               chunks, l.cout, (double)g.mz, stats, buffers + l.rmean_off, buffers + l.rvar_off);
        } else {
            hipLaunchKernelGGL(bn_eval_stats_mb_kernel, dim3((l.cout + 255) / 256), dim3(256), 0, stream,
                               (const float*)(buffers + l.rmean_off), (const float*)(buffers + l.rvar_off), l.cout, stats);
        }
        const float* res = l.res_src >= 0 ? c.ws + c.p.y[l.res_src] : nullptr;
        float* ypre = l.pool ? c.ws + c.p.yp[k] : c.ws + c.p.y[k];
        const long total = g.mz * l.cout;
        {
            HowlProfScope prof("mb_sweep", stream, (res != nullptr ? 12.0 : 8.0) * (double)total);   // z (+ residual) in, y out
            hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(flat_grid(total)), dim3(256), 0, stream, (const float*)z,
                               (const float*)stats, params + l.gamma_off, params + l.beta_off, res, l.act, l.cout, total, ypre);
        }
        if (l.pool) {
            const long tp = g.my * l.cout;
            hipLaunchKernelGGL(maxpool12_fwd_kernel, dim3(flat_grid(tp)), dim3(256), 0, stream, (const float*)ypre, g.wo, g.wy,
                               l.cout, tp, c.ws + c.p.y[k]);
        }
        in = c.ws + c.p.y[k];
        isb = (long)g.hy * g.wy * l.cout;
        ish = (long)g.wy * l.cout;
        isw = l.cout;
        isc = 1;
    }
    const Geo& gl = c.p.g[nl - 1];
    const long tp = (long)B * MB_LAST;
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(flat_grid(tp)), dim3(256), 0, stream, (const float*)in, gl.hy * gl.wy, MB_LAST,
                       training ? drop_mask : (const float*)nullptr, drop_scale, tp, c.ws + c.p.pooled, c.ws + c.p.pooled_d);
    const float* wc = params + c.n->feature_params;
    const float* bc = wc + (size_t)num_labels * MB_LAST;
    gemm(stream, true, c.ws + c.p.pooled_d, lin(MB_LAST), 1, lin(0), wc, lin(1), MB_LAST, B, num_labels, MB_LAST, 1, bc, 0, logits,
         num_labels, 0);
    HOWL_CHECK_LAUNCH("howl_mobilenet_fwd");
    return HOWL_OK;
}

int howl_mobilenet_bwd(const float* params, int num_labels, const float* x, long sb, long sm, long st, int B, int M, int T,
                       const float* drop_mask, float drop_scale, const float* dlogits, float* grads, void* ws, size_t ws_bytes,
                       hipStream_t stream) {
    HOWL_REQUIRE(params && x && dlogits && grads && ws, "howl_mobilenet_bwd: null pointer");
    HOWL_REQUIRE(B >= 1 && M >= 1 && T >= 1 && num_labels >= 1, "howl_mobilenet_bwd: bad shape");
    HOWL_REQUIRE(ws_bytes >= howl_mobilenet_workspace_bytes(B, M, T, num_labels), "howl_mobilenet_bwd: workspace too small");
    Ctx c{&net(), make_plan(B, M, T, num_labels), reinterpret_cast<float*>(ws), params, stream, B};
    const int nl = (int)c.n->layers.size();
    double* part = reinterpret_cast<double*>(c.ws + c.p.part);
    float* scratch = c.ws + c.p.gemm_scratch;
    float* m12 = c.ws + c.p.m12;
    // classifier: logits = pooled_d W^T + b
    const float* wc = params + c.n->feature_params;
    float* gwc = grads + c.n->feature_params;
    float* gbc = gwc + (size_t)num_labels * MB_LAST;
    wgrad_gemm(stream, dlogits, lin(num_labels), num_labels, c.ws + c.p.pooled_d, lin(MB_LAST), MB_LAST, B, scratch, gwc);
    colsum(stream, dlogits, lin(num_labels), B, num_labels, scratch, gbc, nullptr);
    float* dpooled = c.ws + c.p.dpooled;
    gemm(stream, true, dlogits, lin(num_labels), 1, lin(0), wc, lin(MB_LAST), 1, B, MB_LAST, num_labels, 1, nullptr, 0, dpooled,
         MB_LAST, 0);
    {
        const Geo& gl = c.p.g[nl - 1];
        const long total = gl.my * MB_LAST;
        hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(flat_grid(total)), dim3(256), 0, stream, (const float*)dpooled, gl.hy * gl.wy,
                           MB_LAST, drop_mask, drop_scale, total, c.ws + c.p.dact[nl - 1]);
    }
    std::vector<char> has_grad(nl, 0);  // dact[k] already holds a residual contribution
    has_grad[nl - 1] = 1;
    // Two HIP queues (cf. howl_res8_bwd): the BatchNorm-backward -> data-gradient chain stays on the caller's stream; a
    // layer's weight gradient (split-K GEMM / depthwise reduction + slab sums, im2col for the dense layers) only hangs
    // off dz_k and goes to the side queue, where its launches overlap the chain's -- at ~740 short launches per step the
    // backward pass is bound by launch latency, not by the device.  dz is double-buffered (layer k-2 reuses layer k's
    // buffer and waits for wgrad_k); events 0/1: dz ready, 2/3: wgrad done with that dz buffer, 4: join.
    HowlSideQueue* sq = howl_side_queue(stream, 1, "HOWL_MOBILENET_BWD_QUEUES");
    hipStream_t wst = sq ? sq->stream : stream;
    int par = 0, forked = 0;
    for (int k = nl - 1; k >= 0; --k) {
        const HowlMbLayer& l = c.n->layers[k];
        const Geo& g = c.p.g[k];
        const float* z = c.ws + c.p.z[k];
        const float* stats = c.ws + c.p.stats[k];
        const float* dy = c.ws + c.p.dact[k];
        const long total = g.mz * l.cout;
        if (l.pool) {
            float* dyp = c.ws + c.p.dyp;
            hipLaunchKernelGGL(maxpool12_bwd_kernel, dim3(flat_grid(total)), dim3(256), 0, stream,
                               (const float*)(c.ws + c.p.yp[k]), dy, g.wo, g.wy, l.cout, total, dyp);
            dy = dyp;
        }
        if (l.res_src >= 0) {
            // y_k = bn(z_k) + y_src: the same gradient also reaches the block input
            float* dsrc = c.ws + c.p.dact[l.res_src];
            hipMemcpyAsync(dsrc, dy, (size_t)total * sizeof(float), hipMemcpyDeviceToDevice, stream);
            has_grad[l.res_src] = 1;
        }
        const int chunks = chunks_for(g.mz, l.cout);
        const long rpc = (g.mz + chunks - 1) / chunks;
        {
            HowlProfScope prof("mb_sweep", stream, 8.0 * (double)total);                 // z and dy in
            hipLaunchKernelGGL(col_reduce_kernel<1>, dim3((l.cout + 63) / 64, chunks), dim3(256), 0, stream, z, dy, stats,
                               params + l.gamma_off, params + l.beta_off, l.act, g.mz, l.cout, col_pack(l.cout), rpc, part);
        }
        hipLaunchKernelGGL(bn_bwd_finalize_mb_kernel, dim3((l.cout + 63) / 64), dim3(FIN_THREADS), 0, stream, (const double*)part,
                           chunks, l.cout, (double)g.mz, grads + l.gamma_off, grads + l.beta_off, m12);
        float* dz = c.ws + (par ? c.p.dz2 : c.p.dz);
        if (sq && forked >= 2) hipStreamWaitEvent(stream, sq->ev[2 + par], 0);
        {
            HowlProfScope prof("mb_sweep", stream, 12.0 * (double)total);                // z and dy in, dz out
            hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(flat_grid(total)), dim3(256), 0, stream, z, dy, stats,
                               params + l.gamma_off, params + l.beta_off, (const float*)m12, l.act, l.cout, total, dz);
        }
        if (sq) {
            hipEventRecord(sq->ev[par], stream);
            hipStreamWaitEvent(wst, sq->ev[par], 0);
        }
        // convolution backward
        const float* in = k > 0 ? c.ws + c.p.y[k - 1] : x;
        const float* w = params + l.w_off;
        float* gw = grads + l.w_off;
        float* dx = nullptr;
        float* dtmp = nullptr;
        if (k > 0) {
            // the data gradient is written straight into dact[k-1] unless a residual gradient already sits there
            dx = has_grad[k - 1] ? c.ws + c.p.dyp : c.ws + c.p.dact[k - 1];
            dtmp = has_grad[k - 1] ? dx : nullptr;
        }
        const long in_total = (long)B * g.hin * g.win * l.cin;
        if (l.kind == MB_PW) {
            wgrad_gemm(wst, dz, lin(l.cout), l.cout, in, lin(l.cin), l.cin, (int)g.mz, scratch, gw, MB_WGRAD_SPLITS);
            if (dx != nullptr)
                gemm(stream, true, dz, lin(l.cout), 1, lin(0), w, lin(l.cin), 1, (int)g.mz, l.cin, l.cout, 1, nullptr, 0, dx, l.cin,
                     0);
        } else if (l.kind == MB_DW) {
            // chunks of image rows (b, oh): as many as the pixel-chunk rule would give, at least one row each
            const int nrows = B * g.ho;
            int wch = chunks_for(g.mz, 64);
            if (wch > nrows) wch = nrows;
            const int wrpc = (nrows + wch - 1) / wch;
            wch = (nrows + wrpc - 1) / wrpc;
            if (l.stride == 1)
                hipLaunchKernelGGL(dw3x3_wgrad_kernel<1>, dim3((l.cout + 63) / 64, wch), dim3(256), 0, wst, (const float*)dz, in,
                                   g.hin, g.win, l.cin, g.ho, g.wo, nrows, wrpc, scratch);
            else
                hipLaunchKernelGGL(dw3x3_wgrad_kernel<2>, dim3((l.cout + 63) / 64, wch), dim3(256), 0, wst, (const float*)dz, in,
                                   g.hin, g.win, l.cin, g.ho, g.wo, nrows, wrpc, scratch);
            const long nw = (long)l.cout * 9;
            hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)((nw + 63) / 64)), dim3(256), 0, wst, (const float*)scratch,
                               wch, nw, gw);
            if (dx != nullptr) {
                const dim3 grid((unsigned)(B * g.hin), (((g.win + DW_SEG - 1) / DW_SEG) * l.cin + 255) / 256);
                if (l.stride == 1)
                    hipLaunchKernelGGL(dw3x3_dgrad_kernel<1>, grid, dim3(256), 0, stream, (const float*)dz, w, g.hin, g.win, l.cin,
                                       g.ho, g.wo, dx);
                else
                    hipLaunchKernelGGL(dw3x3_dgrad_kernel<2>, grid, dim3(256), 0, stream, (const float*)dz, w, g.hin, g.win, l.cin,
                                       g.ho, g.wo, dx);
            }
        } else {
            float* col = c.ws + c.p.col;
            const int K = 9 * l.cin;
            const long ctot = g.mz * K;
            long isb, ish, isw, isc;
            if (k == 0) {
                isb = sb, ish = sm, isw = st, isc = 0;
            } else {
                isb = (long)g.hin * g.win * l.cin, ish = (long)g.win * l.cin, isw = l.cin, isc = 1;
            }
            hipLaunchKernelGGL(im2col3x3_kernel, dim3(flat_grid(ctot)), dim3(256), 0, wst, in, isb, ish, isw, isc, g.hin, g.win,
                               l.cin, g.ho, g.wo, l.stride, l.pad_h, l.pad_w, ctot, col);
            wgrad_gemm(wst, dz, lin(l.cout), l.cout, col, lin(K), K, (int)g.mz, scratch, gw, MB_WGRAD_SPLITS);
            if (l.bias) {   // sum over pixels of dz (rounding noise in exact arithmetic: a BatchNorm follows the bias);
                            // `part` belongs to the chain, so this reduction stays on the caller's stream
                hipLaunchKernelGGL(col_reduce_kernel<0>, dim3((l.cout + 63) / 64, chunks), dim3(256), 0, stream, (const float*)dz,
                                   (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                                   0, g.mz, l.cout, col_pack(l.cout), rpc, part);
                hipLaunchKernelGGL(colsum_finalize_kernel, dim3((l.cout + 63) / 64), dim3(FIN_THREADS), 0, stream, (const double*)part,
                                   chunks, l.cout, grads + l.b_off);
            }
            if (dx != nullptr) {
                float* dcol = c.ws + c.p.dcol;
                gemm(stream, true, dz, lin(l.cout), 1, lin(0), w, lin(K), 1, (int)g.mz, K, l.cout, 1, nullptr, 0, dcol, K, 0);
                hipLaunchKernelGGL(col2im3x3_kernel, dim3(flat_grid(in_total)), dim3(256), 0, stream, (const float*)dcol, g.hin,
                                   g.win, l.cin, g.ho, g.wo, l.stride, l.pad_h, l.pad_w, in_total, dx);
            }
        }
        if (dtmp != nullptr)
            hipLaunchKernelGGL(add_inplace_kernel, dim3(flat_grid(in_total)), dim3(256), 0, stream, c.ws + c.p.dact[k - 1],
                               (const float*)dtmp, in_total);
        if (sq) hipEventRecord(sq->ev[2 + par], wst);   // this layer's weight gradient no longer needs dz
        par ^= 1;
        ++forked;
    }
    if (sq) {   // join: everything after this call on the caller's stream sees all weight gradients
        hipEventRecord(sq->ev[4], wst);
        hipStreamWaitEvent(stream, sq->ev[4], 0);
    }
    HOWL_CHECK_LAUNCH("howl_mobilenet_bwd");
    return HOWL_OK;
}

}  // extern "C"
