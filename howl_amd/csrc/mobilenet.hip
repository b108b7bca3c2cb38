// MobileNetClassifier ("mobilenet"; reference howl/model/cnn.py:15-29, BASELINE configs[4]): forward and backward.
//
//   downsample = Conv2d(1,3,3,padding=(1,3)) + BatchNorm2d(3) + ReLU + MaxPool2d((1,2))          (cnn.py:18-21)
//   model      = torchvision mobilenet_v2 with a num_labels classifier                           (cnn.py:22-24)
//   forward(x) = model(downsample(x[:, :1]))                                                     (cnn.py:26-29)
// torchvision is a third-party dependency that is not under /root/reference; the body follows the published
// architecture (Sandler et al. 2018 table 2 == torchvision mobilenetv2.py, width 1.0): see oracle/mobilenet.py.
//
// Design.  Activations are channels-last, one row per pixel: tensor k is an (M_k = B*H_k*W_k) x C_k row-major matrix.
// At batch 512 the whole network is ~0.7 GB of activations in 53 layers, most of them a few MB: what a layer costs is the
// number of launches and dependent memory round trips it takes, not bytes.  So a layer is ONE launch forward and ONE backward:
//   * every convolution kernel also reduces the BatchNorm statistics of what it writes (per-channel partial sums -> the last
//     block to finish folds them in a fixed order: `Arrive`), and no BatchNorm / ReLU6 / residual pass exists as a kernel of
//     its own -- the consumer of z_k applies y_k = act(z_k * scale + shift) (+ residual) while it loads its operand
//     (ss_k = [scale | shift | mean | rstd]); only the narrow linear-bottleneck outputs are also stored, by that consumer;
//   * backward: the kernel that produces layer j's incoming gradient multiplies it by the activation mask, stores it (g_j)
//     and reduces sum g, sum g*xhat; dz_j = scale*g_j + c1*z_j + c0 (bc_j) is rebuilt on load by the data- and weight-
//     gradient blocks of layer j, which share one launch; weight-gradient slabs of all layers are folded by one launch;
//   * 1x1 convolutions are fp32 MFMA tile products (64 x 64 x 64 steps); depthwise 3x3 kernels keep lane = channel; the two
//     stem convolutions (1->3, 3->32) are direct, one thread per pixel;
//   * the layer table is built once on the host and published through howl_mobilenet_layer(): the Python module lays
//     its parameters out in ONE flat buffer at those offsets, in PyTorch's own shapes, so gradients land in a flat
//     buffer of the same layout and the optimiser is a single fused AdamW launch.
// Saved for the backward pass (in the caller's workspace): each layer's convolution output z_k, ss_k, the stored y_k of the
// linear bottlenecks; masks and normalised values are recomputed from z_k.  Reductions are fixed-order: results do not
// depend on scheduling.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
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

constexpr int MB_G = 32;          // blocks of one first-level group of a fused per-channel reduction
constexpr int MB_R2 = 64;         // groups (second level); a reduction has at most MB_R = MB_G * MB_R2 blocks along the rows
constexpr int MB_R = MB_G * MB_R2;
constexpr int MB_MAXC = 1280;     // widest layer
constexpr int MB_CBLOCKS = 64;    // channel blocks of a launch: 64 wide (<= 20) or 32 wide (<= 40)
constexpr int MB_COUNTERS = MB_CBLOCKS * (MB_R2 + 1);   // arrival counters: per channel block, one per group + one for the groups
constexpr int MB_MAX_JOBS = 60;   // deferred slab sums of one backward call (53 convolution weights)
constexpr int DW_SEG = 4;

// y_k exists in memory only where something other than one convolution reads it: the linear bottleneck outputs (narrow;
// residual sources and sums).  Every other layer's BatchNorm + activation (+ the downsample's max pool) is applied by its
// consumer while it loads z_k.
inline bool materialized(const HowlMbLayer& l) { return l.act == MB_ACT_NONE; }

struct Plan {
    std::vector<Geo> g;
    std::vector<size_t> z, y, gr, ss, bc, slab;  // float offsets (y: 0 when not materialised)
    std::vector<int> nslab;                          // slabs of the layer's weight gradient
    size_t dz = 0, slab0b = 0, part = 0, part2 = 0, counters = 0, head_scratch = 0, bias_scratch = 0, pooled = 0,
           pooled_d = 0;
    size_t total_floats = 0;
};

// row tiles of a pointwise launch: each block takes a run of consecutive 64-row tiles
inline void pw_rows(long M, int col_tiles, int* tiles_per_block, int* blocks, int tile = 64) {
    const int row_tiles = (int)((M + tile - 1) / tile);
    int tpb = (int)(((long)row_tiles * col_tiles + 4095) / 4096);     // ~4096 blocks at most ...
    if (tpb < (row_tiles + MB_R - 1) / MB_R) tpb = (row_tiles + MB_R - 1) / MB_R;   // ... and <= MB_R of them along the rows
    if (tpb < 1) tpb = 1;
    *tiles_per_block = tpb;
    *blocks = (row_tiles + tpb - 1) / tpb;
}
// tile edge of a pointwise product with an (M x n) result: 32 when 64 x 64 tiles would not give every CU a block
inline int env_int(const char* name, int fallback) {
    const char* v = getenv(name);
    return v != nullptr && v[0] != 0 ? atoi(v) : fallback;
}
inline int pw_tile(long M, int n) {
    const long tiles64 = ((M + 63) / 64) * ((n + 63) / 64);
    static const int per_cu = env_int("HOWL_MB_TILE_BLOCKS_PER_CU", 1);    // fewer 64 x 64 tiles than this per CU -> 32-row tiles
    return tiles64 < (long)per_cu * howl_num_cus() ? 32 : 64;
}
// The weight-gradient blocks of a pointwise backward launch run beside its data-gradient blocks: a split is sized so that both
// kinds take about the same number of 64-deep steps (data gradient: tiles_per_block x ceil(N / 64)), which keeps the slabs
// that the deferred sum has to read few.
inline int pw_wgrad_rows_per_split(long rows, int n, int c) {
    int tpb, blocks;
    pw_rows(rows, (c + 63) / 64, &tpb, &blocks);
    long rps = 64L * tpb * ((n + 63) / 64);
    if (rps < 256) rps = 256;     // (128 .. 256 rows measured the same; 512 and more make these blocks the long pole)
    if (rps > rows) rps = (rows + 63) / 64 * 64;
    return (int)rps;
}
// blocks of the stem's weight-gradient reductions: whole multiples of 256 pixels each, at most 1024 blocks
inline int stem_px_per_block(long pixels) {
    long per = (pixels + 1023) / 1024;
    per = (per + 255) / 256 * 256;
    return (int)(per < 256 ? 256 : per);
}
inline int stem_blocks(long pixels) {
    const long per = stem_px_per_block(pixels);
    return (int)((pixels + per - 1) / per);
}
// chunks of image rows for the depthwise kernels / pixel rows for the narrow column reductions: >= `min_per` units each
inline int row_chunks(long units, int min_per, int* per_chunk) {
    long c = units / min_per;
    c = c < 1 ? 1 : (c > MB_R ? MB_R : c);
    const long per = (units + c - 1) / c;
    *per_chunk = (int)per;
    return (int)((units + per - 1) / per);
}
// Narrow column reductions (stem): a wave covers 64 / Cp rows at a time (Cp = C rounded up to a power of two <= 32)
inline int col_pack(int C) {
    if (C > 32) return 64;
    int cp = 1;
    while (cp < C) cp <<= 1;
    return cp;
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
        p.gr.push_back(take((size_t)g.mz * l.cout));
        p.y.push_back(materialized(l) ? take((size_t)g.my * l.cout) : 0);
        p.ss.push_back(take(4 * (size_t)l.cout));
        p.bc.push_back(take(4 * (size_t)l.cout));
        int ns;
        size_t wn;
        if (l.kind == MB_PW) {
            const int rps = pw_wgrad_rows_per_split(g.mz, l.cout, l.cin);
            ns = (int)((g.mz + rps - 1) / rps);
            wn = (size_t)l.cout * l.cin;
        } else if (l.kind == MB_DW) {
            int per;
            ns = row_chunks((long)B * g.ho, 8, &per);
            wn = (size_t)l.cout * 9;
        } else {
            ns = stem_blocks(g.mz);
            wn = (size_t)l.cout * l.cin * 9;
        }
        p.nslab.push_back(ns);
        p.slab.push_back(take((size_t)ns * wn));
        h = g.hy;
        w = g.wy;
    }
    p.dz = take((size_t)p.g[1].mz * n.layers[1].cout);     // materialised dz of features[0]
    p.slab0b = take((size_t)p.nslab[0] * n.layers[0].cout);  // conv-bias gradient slabs of the downsample layer
    p.part = take((size_t)MB_R * 2 * MB_MAXC);
    p.part2 = take((size_t)MB_R2 * 2 * MB_MAXC);
    p.counters = take(MB_COUNTERS);
    p.head_scratch = take((size_t)64 * num_labels * MB_LAST);
    p.bias_scratch = take((size_t)64 * std::max(num_labels, 64));
    p.pooled = take((size_t)B * MB_LAST);
    p.pooled_d = take((size_t)B * MB_LAST);
    p.total_floats = off;
    return p;
}

// ---------------------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int chan_of(long idx, int C) {
    return idx < (1L << 31) ? (int)((unsigned)idx % (unsigned)C) : (int)(idx % C);
}
__device__ __forceinline__ bool mb_act_passes(float v, int act) {  // derivative of the activation is 1
    if (act == MB_ACT_RELU6) return v > 0.0f && v < 6.0f;
    if (act == MB_ACT_RELU) return v > 0.0f;
    return true;
}
__device__ __forceinline__ float relu6f(float v) { return fminf(fmaxf(v, 0.0f), 6.0f); }

// BatchNorm in the form every kernel uses: y = act(z * scale + shift), scale = gamma * rstd, shift = beta - mean * scale
// (ss_k = [scale | shift | mean | rstd], 4 C floats); the SAME fmaf decides the ReLU6 mask in the backward pass.
// Backward: with g = dy * act'(.), m1 = mean(g), m2 = mean(g * xhat):
//   dz = scale * (g - m1 - xhat * m2) = scale * g + c1 * z + c0,  c1 = -scale * m2 * rstd,  c0 = -scale * m1 - c1 * mean
// (bc_k = [scale | c1 | c0]) so that the consumers of dz_k rebuild it from g_k and z_k with two fmaf.
struct FinFwd {
    const float* gamma;
    const float* beta;
    float* rmean;
    float* rvar;
    float* ss;
    double count;
};
struct FinBwd {
    const float* ss;
    float* dgamma;
    float* dbeta;
    float* bc;
    double count;
};

// ---- "the last block to finish folds everybody's partials", without a device-wide fence ---------------------------------
// A release fence at device scope writes back the whole L2 of the XCD (measured: ~90 ns per block, serialised, when every
// thread fences; tools/fence_probe.hip).  Instead the few floats that cross blocks INSIDE a launch -- one row of per-channel
// partial sums per block -- are written with device-scope relaxed atomic stores (write-through), the storing thread waits
// for them (vmcnt(0)), the block meets at a barrier, and thread 0 bumps an arrival counter with a relaxed device-scope
// atomic; the block that draws the last ticket reads the rows back with device-scope atomic loads.  Everything else a
// kernel writes is only read by later launches.  Two levels keep every fold to one round trip to memory (~1.5 us): the last
// block of each group of MB_G blocks folds the group's rows into one row, the last group folds the <= MB_R2 group rows and
// finalises.  Rows are folded in index order, so WHICH block is last does not change a bit of the result; counters are
// left at zero for the next launch in stream order.
struct Arrive {
    float* part1;      // [blocks along rows][2][C]
    float* part2;      // [groups][2][C]
    unsigned* cnt1;    // [channel block][MB_R2]
    unsigned* cnt2;    // [channel block]
};
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// `overlap()` runs in every thread while thread 0's arrival is on its way (~2 us to the device-coherent level and back): the
// pointwise kernels issue the stores of their last output tile there, so the arrival does not wait for them.
struct NoOverlap {
    __device__ __forceinline__ void operator()() const {}
};
// The hand-off above is a gfx950 protocol, not HIP memory-model code: it relies on device-scope atomic stores being
// write-through to the device-coherent level, on vmcnt(0) meaning "acknowledged there", and on the gfx9 encoding of s_waitcnt
// (simm16 0x0F70 = vmcnt(0), expcnt / lgkmcnt unconstrained).  A release / acquire pair on the counter would be portable and
// costs a write-back of the XCD's L2 per block (tools/fence_probe.hip: 28-90 ns per block, serialised) -- unaffordable here.
// This library is built for gfx950 only; any other target must not compile this silently.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "mobilenet.hip: the fence-free arrival protocol is validated on gfx950 only (see the comment above arrive())"
#endif
constexpr int HOWL_GFX9_WAIT_VMCNT0 = 0x0F70;
template <class Overlap>
__device__ __forceinline__ bool arrive(unsigned* counter, unsigned expected, const Overlap& overlap) {
    __shared__ unsigned ticket;
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    __builtin_amdgcn_s_waitcnt(HOWL_GFX9_WAIT_VMCNT0);   // vmcnt(0): this thread's write-through stores are visible device-wide
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == expected - 1) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = t;
    }
    overlap();
    __syncthreads();
    return ticket == expected - 1;
}

// Block `by` of the R1 blocks that share channel block `cb` (cw channels from cw * cb); the thread with `own` holds this
// block's (v0, v1) of column c_own.  Returns true in the one block of the channel block that must finalise; a.part2 then
// holds all groups' rows.
template <class Overlap = NoOverlap>
__device__ __forceinline__ bool publish_and_arrive(const Arrive& a, int C, int cb, int by, int R1, bool own, int c_own, float v0,
                                                   float v1, const Overlap& overlap = Overlap(), int cw = 64) {
    __shared__ float red[2][4][64];
    const int c0 = cb * cw;     // channel blocks are cw <= 64 channels wide (lanes >= cw idle in the folds)
    // groups of G blocks, G the smallest size that leaves <= MB_R2 groups: launches with few blocks along the rows (most of
    // this network) have G = 1 and skip the first level altogether -- their rows ARE the group rows
    const int G = (R1 + MB_R2 - 1) / MB_R2;
    const int R2 = (R1 + G - 1) / G;
    if (G == 1) {
        if (own) {
            st_agent(&a.part2[((size_t)by * 2 + 0) * C + c_own], v0);
            st_agent(&a.part2[((size_t)by * 2 + 1) * C + c_own], v1);
        }
        return arrive(a.cnt2 + cb, R2, overlap);
    }
    if (own) {
        st_agent(&a.part1[((size_t)by * 2 + 0) * C + c_own], v0);
        st_agent(&a.part1[((size_t)by * 2 + 1) * C + c_own], v1);
    }
    const int group = by / G, g0 = group * G;
    const int gsize = R1 - g0 < G ? R1 - g0 : G;
    if (!arrive(a.cnt1 + cb * MB_R2 + group, gsize, overlap)) return false;
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = c0 + lane < C ? c0 + lane : C - 1;
    float r0[MB_G / 4], r1[MB_G / 4];
#pragma unroll
    for (int u = 0; u < MB_G / 4; ++u) {     // rows g0 + rg, g0 + rg + 4, ...: all loads in flight together
        const int k = g0 + rg + 4 * u < g0 + gsize ? g0 + rg + 4 * u : g0 + gsize - 1;
        r0[u] = ld_agent(&a.part1[((size_t)k * 2 + 0) * C + c]);
        r1[u] = ld_agent(&a.part1[((size_t)k * 2 + 1) * C + c]);
    }
    float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
    for (int u = 0; u < MB_G / 4; ++u) {
        const bool ok = rg + 4 * u < gsize;
        t0 += ok ? r0[u] : 0.0f;
        t1 += ok ? r1[u] : 0.0f;
    }
    red[0][rg][lane] = t0;
    red[1][rg][lane] = t1;
    __syncthreads();
    if (rg == 0 && lane < cw && c0 + lane < C) {
        st_agent(&a.part2[((size_t)group * 2 + 0) * C + c], ((red[0][0][lane] + red[0][1][lane]) + red[0][2][lane]) + red[0][3][lane]);
        st_agent(&a.part2[((size_t)group * 2 + 1) * C + c], ((red[1][0][lane] + red[1][1][lane]) + red[1][2][lane]) + red[1][3][lane]);
    }
    return arrive(a.cnt2 + cb, R2, NoOverlap());
}

__device__ __forceinline__ int arrive_rows(int R1) {   // rows of part2 after publish_and_arrive over R1 blocks
    const int G = (R1 + MB_R2 - 1) / MB_R2;
    return (R1 + G - 1) / G;
}

// part[k][0..1][C] (k < R <= MB_R2) -> column totals of column c; called by all 256 threads of the finalising block, lane =
// column, the 4 waves take k = wave, wave + 4, ... (all loads in flight together), combined in a fixed order in fp64.
__device__ __forceinline__ void fold_partials(const float* part, int R, int C, int c, double& s0, double& s1) {
    __shared__ double red[2][4][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int cc = c < C ? c : C - 1;
    float v0[MB_R2 / 4], v1[MB_R2 / 4];
#pragma unroll
    for (int u = 0; u < MB_R2 / 4; ++u) {
        const int k = rg + 4 * u < R ? rg + 4 * u : R - 1;
        v0[u] = ld_agent(&part[((size_t)k * 2 + 0) * C + cc]);
        v1[u] = ld_agent(&part[((size_t)k * 2 + 1) * C + cc]);
    }
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int u = 0; u < MB_R2 / 4; ++u) {
        const bool ok = rg + 4 * u < R;
        a0 += ok ? (double)v0[u] : 0.0;
        a1 += ok ? (double)v1[u] : 0.0;
    }
    red[0][rg][lane] = a0;
    red[1][rg][lane] = a1;
    __syncthreads();
    s0 = ((red[0][0][lane] + red[0][1][lane]) + red[0][2][lane]) + red[0][3][lane];
    s1 = ((red[1][0][lane] + red[1][1][lane]) + red[1][2][lane]) + red[1][3][lane];
}

// batch statistics of column c (biased variance for normalisation, unbiased for the running estimate: nn.BatchNorm2d)
__device__ __forceinline__ void finalize_fwd(const float* part, int R, int C, int c, const FinFwd& f) {
    double s0, s1;
    fold_partials(part, R, C, c, s0, s1);
    if ((threadIdx.x >> 6) != 0 || c >= C) return;
    const double mean = s0 / f.count;
    double var = s1 / f.count - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const double rstd = 1.0 / sqrt(var + (double)MB_EPS);
    const double scale = (double)f.gamma[c] * rstd;
    f.ss[c] = (float)scale;
    f.ss[C + c] = (float)((double)f.beta[c] - mean * scale);
    f.ss[2 * C + c] = (float)mean;
    f.ss[3 * C + c] = (float)rstd;
    const double unbiased = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
    f.rmean[c] = (float)((1.0 - MB_MOMENTUM) * (double)f.rmean[c] + MB_MOMENTUM * mean);
    f.rvar[c] = (float)((1.0 - MB_MOMENTUM) * (double)f.rvar[c] + MB_MOMENTUM * unbiased);
}

// BatchNorm backward of column c: dbeta = sum g, dgamma = sum g * xhat, and the coefficients of dz (see above)
__device__ __forceinline__ void finalize_bwd(const float* part, int R, int C, int c, const FinBwd& f) {
    double s0, s1;
    fold_partials(part, R, C, c, s0, s1);
    if ((threadIdx.x >> 6) != 0 || c >= C) return;
    f.dbeta[c] = (float)s0;
    f.dgamma[c] = (float)s1;
    const double scale = (double)f.ss[c], mean = (double)f.ss[2 * C + c], rstd = (double)f.ss[3 * C + c];
    const double m1 = s0 / f.count, m2 = s1 / f.count;
    const double c1 = -scale * m2 * rstd;
    f.bc[c] = (float)scale;
    f.bc[C + c] = (float)c1;
    f.bc[2 * C + c] = (float)(-scale * m1 - c1 * mean);
}

// ---------------------------------------------------------------------------------------------------------
// Pointwise (1x1) convolutions.  Shared tile engine: a (32 WTR) x (32 WTC) block of outputs per step, 64 deep (PK), 2x2 waves
// x (WTR x WTC) fp32 MFMA 16x16x4 tiles; the operand pieces of the NEXT step are requested (16-byte loads) before the MFMAs of
// the current one (two steps in flight was measured: slower, the registers cost occupancy).
// The late layers of this network are small matrices with a deep reduction (2048 x 960 x 160).  On 64 x 64 tiles that is 96
// blocks for 256 CUs, each a chain of 15 steps of 64 MFMAs per wave (32 cycles each): MFMA-bound per block on a mostly idle
// chip.  Launches with fewer 64 x 64 tiles than CUs take 32-row tiles instead (forward 32 x 32, data gradient 32 x 64: its A
// operand is two tensors, re-reading it per column block costs more than it buys) -- shorter chains on more CUs.
// A thread's piece keeps ONE k offset for the whole launch: the per-channel constants of the on-load transforms are a few
// registers per step.
// Operand tiles whose unit-stride index is the reduction index are kept [row][k] (stride LDK): 16-byte stores, and a lane
// group q = lane / 16 reads k = 16t + 4q .. + 3 with ONE 16-byte LDS load and feeds them to four consecutive MFMAs (the order
// in which the 64 k of a step enter the sum is a fixed permutation).  Tiles whose unit stride is the output column stay
// [k][col] (stride LDC).
// ---------------------------------------------------------------------------------------------------------
constexpr int PK = 64;
constexpr int LDK = PK + 4;   // floats: 16-byte aligned rows, conflict-free fragment loads
constexpr int LDC = 80;
constexpr int PW_LDS_FLOATS = 2 * PK * LDC;   // the largest role (weight gradient: two [k][col] tiles)

enum { PW_IN_PLAIN = 0, PW_IN_RELU6 = 1, PW_IN_LINEAR = 2 };
__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 bn_lin_4(float4 v, float4 sc, float4 sh) {
    return make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
}
__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// T(v) = relu6(v * scale + shift), four channels
__device__ __forceinline__ float4 bn_relu6_4(float4 v, float4 sc, float4 sh) {
    v.x = relu6f(fmaf(v.x, sc.x, sh.x));
    v.y = relu6f(fmaf(v.y, sc.y, sh.y));
    v.z = relu6f(fmaf(v.z, sc.z, sh.z));
    v.w = relu6f(fmaf(v.w, sc.w, sh.w));
    return v;
}
// dz = scale * g + c1 * z + c0, four channels
__device__ __forceinline__ float4 dz4(float4 g, float4 z, float4 sc, float4 c1, float4 c0) {
    float4 v;
    v.x = fmaf(sc.x, g.x, fmaf(c1.x, z.x, c0.x));
    v.y = fmaf(sc.y, g.y, fmaf(c1.y, z.y, c0.y));
    v.z = fmaf(sc.z, g.z, fmaf(c1.z, z.z, c0.z));
    v.w = fmaf(sc.w, g.w, fmaf(c1.w, z.w, c0.w));
    return v;
}
__device__ __forceinline__ float f4_get(const float4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); }

// `groups` x 16 k of the step: A and B both [row][k]
template <int WTR, int WTC>
__device__ __forceinline__ void mma_kk(const float* As, const float* Bs, f32x4 (&acc)[WTR][WTC], int groups, int wr,
                                       int wc, int lane) {
    const int q = lane >> 4, l15 = lane & 15;
    for (int t = 0; t < groups; ++t) {
        float4 a[WTR], b[WTC];
#pragma unroll
        for (int i = 0; i < WTR; ++i) a[i] = *reinterpret_cast<const float4*>(&As[(16 * WTR * wr + 16 * i + l15) * LDK + 16 * t + 4 * q]);
#pragma unroll
        for (int j = 0; j < WTC; ++j) b[j] = *reinterpret_cast<const float4*>(&Bs[(16 * WTC * wc + 16 * j + l15) * LDK + 16 * t + 4 * q]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < WTR; ++i)
#pragma unroll
                for (int j = 0; j < WTC; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4_get(a[i], e), f4_get(b[j], e), acc[i][j], 0, 0, 0);
    }
}
// A [row][k], B [k][col]
template <int WTR, int WTC>
__device__ __forceinline__ void mma_kc(const float* As, const float* Bs, f32x4 (&acc)[WTR][WTC], int groups, int wr,
                                       int wc, int lane) {
    const int q = lane >> 4, l15 = lane & 15;
    for (int t = 0; t < groups; ++t) {
        float4 a[WTR];
#pragma unroll
        for (int i = 0; i < WTR; ++i) a[i] = *reinterpret_cast<const float4*>(&As[(16 * WTR * wr + 16 * i + l15) * LDK + 16 * t + 4 * q]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float b[WTC];
#pragma unroll
            for (int j = 0; j < WTC; ++j) b[j] = Bs[(16 * t + 4 * q + e) * LDC + 16 * WTC * wc + 16 * j + l15];
#pragma unroll
            for (int i = 0; i < WTR; ++i)
#pragma unroll
                for (int j = 0; j < WTC; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4_get(a[i], e), b[j], acc[i][j], 0, 0, 0);
        }
    }
}
// A [k][row], B [k][col]
__device__ __forceinline__ void mma_cc(const float* As, const float* Bs, f32x4 (&acc)[2][2], int groups, int wr, int wc,
                                       int lane) {
    const int q = lane >> 4, l15 = lane & 15;
    for (int t = 0; t < groups; ++t) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[(16 * t + 4 * q + e) * LDC + 32 * wr + 16 * i + l15];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[(16 * t + 4 * q + e) * LDC + 32 * wc + 16 * j + l15];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// forward:  z[m][n] = sum_k T(a[m][k]) * w[n][k], the input's BatchNorm applied while the tile is staged:
//   PW_IN_RELU6:  a = z_{k-1}, T(v) = relu6(v * scale[k] + shift[k])                       (input = a depthwise layer)
//   PW_IN_LINEAR: a = z_{k-1}, T(v) = v * scale[k] + shift[k] (+ res[m][k]); T(a) = y_{k-1} is also stored, once, because the
//                 residual sums and the weight gradient read it again                          (input = a linear bottleneck)
//   PW_IN_PLAIN:  a is used as it is.  A block owns 64 output channels and `tiles_per_block` consecutive 64-row
//   tiles; beside z it keeps the per-channel sum and sum of squares of everything it wrote, publishes them as ONE partial
//   row, and the last block of the channel block turns the partials into this layer's ss (and running statistics).
// ---------------------------------------------------------------------------------------------------------
struct PwFwdStage {
    float4 va[4], vb[4], vr[4], xsc, xsh;
};
template <int XF, int WTR, int WTC>
__global__ __launch_bounds__(256) void pw_fwd_kernel(const float* __restrict__ a, const float* __restrict__ ss_in,
                                                     const float* __restrict__ w, const float* __restrict__ res,
                                                     float* __restrict__ y_out, int M, int N, int K, int tiles_per_block,
                                                     float* __restrict__ z, Arrive arr, FinFwd fin) {
    constexpr int TR = 32 * WTR, TC = 32 * WTC;    // block tile: 2 x 2 waves x (WTR x WTC) MFMA tiles
    constexpr int NPA = TR / 16, NPB = TC / 16;    // operand pieces per thread and step
    __shared__ __attribute__((aligned(16))) float lds[(TR + TC) * LDK];
    __shared__ float sred[2][2][TC];
    float* As = lds;
    float* Bs = lds + TR * LDK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int n0 = blockIdx.x * TC;
    const int p_r = tid >> 4, p_k = (tid & 15) * 4;   // pieces: rows p_r + 16 i, 4 consecutive k from p_k
    const float* bp[NPB];
    bool b_ok[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        bp[i] = w + (long)min(n0 + p_r + 16 * i, N - 1) * K;
        b_ok[i] = n0 + p_r + 16 * i < N;
    }
    const int row_tiles = (M + TR - 1) / TR;
    const int t0 = blockIdx.y * tiles_per_block;
    const int t1 = min(row_tiles, t0 + tiles_per_block);
    float cs[WTC], cq[WTC];
#pragma unroll
    for (int j = 0; j < WTC; ++j) cs[j] = cq[j] = 0.0f;
    f32x4 acc[WTR][WTC];     // the last tile's results stay in registers: its stores are issued under the arrival (see `arrive`)
    auto store_tile = [&](int m0) {
#pragma unroll
        for (int i = 0; i < WTR; ++i)
#pragma unroll
            for (int j = 0; j < WTC; ++j) {
                const int n = n0 + 16 * WTC * wc + 16 * j + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + 16 * WTR * wr + 16 * i + 4 * (lane >> 4) + r;
                    if (m < M && n < N) z[(long)m * N + n] = acc[i][j][r];
                }
            }
    };
    const bool stats = arr.part1 != nullptr;
    for (int t = t0; t < t1; ++t) {
        const int m0 = t * TR;
        const float* ap[NPA];
        bool a_ok[NPA];
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            ap[i] = a + (long)min(m0 + p_r + 16 * i, M - 1) * K;
            a_ok[i] = m0 + p_r + 16 * i < M;
        }
        auto fetch = [&](PwFwdStage& sg, int k0) {
            const int k = min(k0 + p_k, K - 4);
#pragma unroll
            for (int i = 0; i < NPA; ++i) sg.va[i] = ldg4(ap[i] + k);
#pragma unroll
            for (int i = 0; i < NPB; ++i) sg.vb[i] = ldg4(bp[i] + k);
            if (XF != PW_IN_PLAIN) {
                sg.xsc = ldg4(ss_in + k);
                sg.xsh = ldg4(ss_in + K + k);
            }
            if (XF == PW_IN_LINEAR && res != nullptr) {
#pragma unroll
                for (int i = 0; i < NPA; ++i) sg.vr[i] = ldg4(res + (ap[i] - a) + k);
            }
        };
        auto stage = [&](const PwFwdStage& sg, int k0) {
            const bool kok = k0 + p_k < K;
#pragma unroll
            for (int i = 0; i < NPA; ++i) {
                float4 v = sg.va[i];
                if (XF == PW_IN_RELU6) v = bn_relu6_4(v, sg.xsc, sg.xsh);
                if (XF == PW_IN_LINEAR) {
                    v = bn_lin_4(v, sg.xsc, sg.xsh);
                    if (res != nullptr) v = add4(v, sg.vr[i]);
                    // y_{k-1} itself is needed again (residual source, weight gradient): the first channel block stores it
                    if (blockIdx.x == 0 && a_ok[i] && kok) *reinterpret_cast<float4*>(y_out + (ap[i] - a) + k0 + p_k) = v;
                }
                if (!(a_ok[i] && kok)) v = f4_zero();
                *reinterpret_cast<float4*>(&As[(p_r + 16 * i) * LDK + p_k]) = v;
            }
#pragma unroll
            for (int i = 0; i < NPB; ++i)
                *reinterpret_cast<float4*>(&Bs[(p_r + 16 * i) * LDK + p_k]) = (b_ok[i] && kok) ? sg.vb[i] : f4_zero();
        };
#pragma unroll
        for (int i = 0; i < WTR; ++i)
#pragma unroll
            for (int j = 0; j < WTC; ++j) acc[i][j] = {0.0f, 0.0f, 0.0f, 0.0f};
        PwFwdStage st0;
        fetch(st0, 0);
        for (int k0 = 0; k0 < K; k0 += PK) {
            stage(st0, k0);
            __syncthreads();
            fetch(st0, k0 + PK);     // past the end: a clamped (re)load that is never staged
            __builtin_amdgcn_sched_barrier(0);
            mma_kk<WTR, WTC>(As, Bs, acc, min(PK, K - k0 + 15) / 16, wr, wc, lane);
            __syncthreads();
        }
        if (!(stats && t == t1 - 1)) store_tile(m0);
#pragma unroll
        for (int i = 0; i < WTR; ++i)
#pragma unroll
            for (int j = 0; j < WTC; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[i][j][r];          // rows >= M and columns >= N are exact zeros
                    cs[j] += v;
                    cq[j] = fmaf(v, v, cq[j]);
                }
    }
    if (!stats) return;
#pragma unroll
    for (int j = 0; j < WTC; ++j) {
        cs[j] += __shfl_xor(cs[j], 16);
        cs[j] += __shfl_xor(cs[j], 32);
        cq[j] += __shfl_xor(cq[j], 16);
        cq[j] += __shfl_xor(cq[j], 32);
        if (lane < 16) {
            sred[0][wr][16 * WTC * wc + 16 * j + lane] = cs[j];
            sred[1][wr][16 * WTC * wc + 16 * j + lane] = cq[j];
        }
    }
    __syncthreads();
    const bool own = tid < TC && n0 + tid < N;
    const float v0 = own ? sred[0][0][tid] + sred[0][1][tid] : 0.0f, v1 = own ? sred[1][0][tid] + sred[1][1][tid] : 0.0f;
    const int m_last = (t1 - 1) * TR;
    if (publish_and_arrive(arr, N, blockIdx.x, blockIdx.y, gridDim.y, own, n0 + tid, v0, v1, [&]() { store_tile(m_last); }, TC))
        finalize_fwd(arr.part2, arrive_rows(gridDim.y), N, lane < TC ? n0 + lane : N, fin);
}

// What the producer of a data gradient does with its result before it leaves the registers: dy_j (+ the gradient that
// reaches y_j through a residual connection) -> g_j = dy_j * act'(.) is stored, and sum g_j, sum g_j * xhat_j per channel
// go to the partials that end in layer j's BatchNorm parameter gradients and bc_j.
struct EpiBwd {
    const float* addend;   // g of the later layer whose output adds y_j, or nullptr
    const float* zj;
    const float* ssj;
    float* gj;
    int act;
};

// Backward of a pointwise layer k, ONE launch: the first d_cx * d_ry blocks compute the data gradient (and reduce the
// BatchNorm backward of layer j = k-1), the rest the weight gradient -- both only need (g_k, z_k, bc_k), and together they
// fill the chip where either alone would not.
struct PwBwd {
    const float* g;        // g_k, z_k: (M x N); bc_k: 3 N
    const float* zk;
    const float* bc;
    const float* w;        // (N x C)
    const float* in;       // (M x C): y_{k-1}, or z_{k-1} with ss_in
    const float* ss_in;
    int M, N, C;
    int tiles_per_block, d_cx, d_ry;        // data gradient: column blocks x row blocks
    int rows_per_split, w_cx, w_ny, w_nz;   // weight gradient: (C / 64) x (N / 64) x splits
    EpiBwd e;
    Arrive arr;
    FinBwd fin;
    float* slabs;
};

// data gradient:  dy_j[m][c] = sum_n dz[m][n] * w[n][c],  dz = bc.scale*g + bc.c1*z + bc.c0 rebuilt while the tile is staged
// (g_k, z_k: two 16-byte loads per piece); epilogue = EpiBwd for layer j (C = its channels).
struct PwBwdStage {
    float4 vg[4], vz[4], vb[4], ksc, kc1, kc0;
};
template <int WTR, int WTC>
__device__ __forceinline__ void pw_dgrad_body(float* lds, const PwBwd& p, int bx, int by) {
    constexpr int TR = 32 * WTR, TC = 32 * WTC;    // block tile
    constexpr int NP = TR / 16;                    // dz pieces per thread and step
    __shared__ float sred[2][2][TC];
    float* As = lds;                 // [m][n]   (TR x LDK)
    float* Bs = lds + TR * LDK;      // [n][c]   (PK x LDC, TC columns used)
    const int M = p.M, N = p.N, C = p.C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int c0 = bx * TC;
    const int p_r = tid >> 4, p_k = (tid & 15) * 4;
    const int bcol = min(c0 + p_k, C - 4);          // weight piece: rows (= n) p_r + 16 i, 4 consecutive c from p_k
    const bool b_col_ok = c0 + p_k < C && p_k < TC;
    float jsc[WTC], jsh[WTC], jme[WTC], jrs[WTC];           // layer j's constants of this thread's two output columns
#pragma unroll
    for (int j = 0; j < WTC; ++j) {
        const int c = min(c0 + 16 * WTC * wc + 16 * j + (lane & 15), C - 1);
        jsc[j] = p.e.ssj[c];
        jsh[j] = p.e.ssj[C + c];
        jme[j] = p.e.ssj[2 * C + c];
        jrs[j] = p.e.ssj[3 * C + c];
    }
    const int row_tiles = (M + TR - 1) / TR;
    const int t0 = by * p.tiles_per_block;
    const int t1 = min(row_tiles, t0 + p.tiles_per_block);
    float s1[WTC], s2[WTC];
#pragma unroll
    for (int j = 0; j < WTC; ++j) s1[j] = s2[j] = 0.0f;
    float gq[WTR][WTC][4];     // g_j of the current tile; the last tile's stores are issued under the arrival
    auto store_tile = [&](int m0) {
#pragma unroll
        for (int i = 0; i < WTR; ++i)
#pragma unroll
            for (int j = 0; j < WTC; ++j) {
                const int c = c0 + 16 * WTC * wc + 16 * j + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + 16 * WTR * wr + 16 * i + 4 * (lane >> 4) + r;
                    if (m < M && c < C) p.e.gj[(long)m * C + c] = gq[i][j][r];
                }
            }
    };
    for (int t = t0; t < t1; ++t) {
        const int m0 = t * TR;
        long arow[NP];
        bool a_ok[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            arow[i] = (long)min(m0 + p_r + 16 * i, M - 1) * N;
            a_ok[i] = m0 + p_r + 16 * i < M;
        }
        auto fetch = [&](PwBwdStage& sg, int k0) {
            const int k = min(k0 + p_k, N - 4);
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                sg.vg[i] = ldg4(p.g + arow[i] + k);
                sg.vz[i] = ldg4(p.zk + arow[i] + k);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) sg.vb[i] = ldg4(p.w + (long)min(k0 + p_r + 16 * i, N - 1) * C + bcol);
            sg.ksc = ldg4(p.bc + k);
            sg.kc1 = ldg4(p.bc + N + k);
            sg.kc0 = ldg4(p.bc + 2 * N + k);
        };
        auto stage = [&](const PwBwdStage& sg, int k0) {
            const bool kok = k0 + p_k < N;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                float4 v = dz4(sg.vg[i], sg.vz[i], sg.ksc, sg.kc1, sg.kc0);
                if (!(a_ok[i] && kok)) v = f4_zero();
                *reinterpret_cast<float4*>(&As[(p_r + 16 * i) * LDK + p_k]) = v;
            }
            if (p_k < TC) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<float4*>(&Bs[(p_r + 16 * i) * LDC + p_k]) =
                        (b_col_ok && k0 + p_r + 16 * i < N) ? sg.vb[i] : f4_zero();
            }
        };
        f32x4 acc[WTR][WTC];
#pragma unroll
        for (int i = 0; i < WTR; ++i)
#pragma unroll
            for (int j = 0; j < WTC; ++j) acc[i][j] = {0.0f, 0.0f, 0.0f, 0.0f};
        PwBwdStage st0;
        fetch(st0, 0);
        for (int k0 = 0; k0 < N; k0 += PK) {
            stage(st0, k0);
            __syncthreads();
            fetch(st0, k0 + PK);
            __builtin_amdgcn_sched_barrier(0);
            mma_kc<WTR, WTC>(As, Bs, acc, min(PK, N - k0 + 15) / 16, wr, wc, lane);
            __syncthreads();
        }
        // epilogue: all loads first (clamped, unconditional), then arithmetic, then the guarded stores
        float zj[WTR][WTC][4], ad[WTR][WTC][4];
#pragma unroll
        for (int i = 0; i < WTR; ++i)
#pragma unroll
            for (int j = 0; j < WTC; ++j) {
                const int c = min(c0 + 16 * WTC * wc + 16 * j + (lane & 15), C - 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = min(m0 + 16 * WTR * wr + 16 * i + 4 * (lane >> 4) + r, M - 1);
                    zj[i][j][r] = p.e.zj[(long)m * C + c];
                    ad[i][j][r] = p.e.addend != nullptr ? p.e.addend[(long)m * C + c] : 0.0f;
                }
            }
#pragma unroll
        for (int i = 0; i < WTR; ++i)
#pragma unroll
            for (int j = 0; j < WTC; ++j) {
                const int c = c0 + 16 * WTC * wc + 16 * j + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + 16 * WTR * wr + 16 * i + 4 * (lane >> 4) + r;
                    const bool ok = m < M && c < C;
                    const float dy = acc[i][j][r] + ad[i][j][r];
                    const float zv = zj[i][j][r];
                    const bool pass = mb_act_passes(fmaf(zv, jsc[j], jsh[j]), p.e.act);
                    const float gg = (ok && pass) ? dy : 0.0f;
                    gq[i][j][r] = gg;
                    s1[j] += gg;
                    s2[j] = fmaf(gg, (zv - jme[j]) * jrs[j], s2[j]);
                }
            }
        if (t != t1 - 1) store_tile(m0);
    }
#pragma unroll
    for (int j = 0; j < WTC; ++j) {
        s1[j] += __shfl_xor(s1[j], 16);
        s1[j] += __shfl_xor(s1[j], 32);
        s2[j] += __shfl_xor(s2[j], 16);
        s2[j] += __shfl_xor(s2[j], 32);
        if (lane < 16) {
            sred[0][wr][16 * WTC * wc + 16 * j + lane] = s1[j];
            sred[1][wr][16 * WTC * wc + 16 * j + lane] = s2[j];
        }
    }
    __syncthreads();
    const bool own = tid < TC && c0 + tid < C;
    const float v0 = own ? sred[0][0][tid] + sred[0][1][tid] : 0.0f, v1 = own ? sred[1][0][tid] + sred[1][1][tid] : 0.0f;
    const int m_last = (t1 - 1) * TR;
    if (publish_and_arrive(p.arr, C, bx, by, p.d_ry, own, c0 + tid, v0, v1, [&]() { store_tile(m_last); }, TC))
        finalize_bwd(p.arr.part2, arrive_rows(p.d_ry), C, lane < TC ? c0 + lane : C, p.fin);
}

// weight gradient:  dW[n][c] = sum_m dz[m][n] * T(in[m][c]) over this split's rows; both operands are rebuilt while they are
// staged (dz from g_k, z_k, bc_k; T as in pw_fwd_kernel).  A thread's pieces are 4 consecutive columns of rows p_r + 16 i, so
// its per-column constants live in registers for the whole launch.  slabs[split][n][c].
template <bool XF>
__device__ __forceinline__ void pw_wgrad_body(float* lds, const PwBwd& p, int bx, int by, int bz) {
    float* As = lds;                 // [m][n]   (PK x LDC)
    float* Bs = lds + PK * LDC;      // [m][c]
    const int M = p.M, N = p.N, C = p.C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int c0 = bx * GT, n0 = by * GT;
    const int p_r = tid >> 4, p_c = (tid & 15) * 4;
    const int an = min(n0 + p_c, N - 4), bcn = min(c0 + p_c, C - 4);
    const bool a_ok = n0 + p_c < N, b_ok = c0 + p_c < C;
    const float4 ksc = ldg4(p.bc + an), kc1 = ldg4(p.bc + N + an), kc0 = ldg4(p.bc + 2 * N + an);
    float4 isc = f4_zero(), ish = f4_zero();
    if (XF) {
        isc = ldg4(p.ss_in + bcn);
        ish = ldg4(p.ss_in + C + bcn);
    }
    const int mbeg = bz * p.rows_per_split;
    const int mend = min(M, mbeg + p.rows_per_split);
    float4 vg[4], vz[4], vb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long m = (long)min(k0 + p_r + 16 * i, M - 1);
            vg[i] = ldg4(p.g + m * N + an);
            vz[i] = ldg4(p.zk + m * N + an);
            vb[i] = ldg4(p.in + m * C + bcn);
        }
    };
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = {0.0f, 0.0f, 0.0f, 0.0f};
    fetch(mbeg);
    for (int k0 = mbeg; k0 < mend; k0 += PK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool rok = k0 + p_r + 16 * i < mend;
            float4 va = dz4(vg[i], vz[i], ksc, kc1, kc0);
            float4 vv = XF ? bn_relu6_4(vb[i], isc, ish) : vb[i];
            if (!(a_ok && rok)) va = f4_zero();
            if (!(b_ok && rok)) vv = f4_zero();
            *reinterpret_cast<float4*>(&As[(p_r + 16 * i) * LDC + p_c]) = va;
            *reinterpret_cast<float4*>(&Bs[(p_r + 16 * i) * LDC + p_c]) = vv;
        }
        __syncthreads();
        fetch(k0 + PK);
        __builtin_amdgcn_sched_barrier(0);
        mma_cc(As, Bs, acc, min(PK, mend - k0 + 15) / 16, wr, wc, lane);
        __syncthreads();
    }
    float* out = p.slabs + (size_t)bz * N * C;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = c0 + 32 * wc + 16 * j + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + 32 * wr + 16 * i + 4 * (lane >> 4) + r;
                if (n < N && c < C) out[(long)n * C + c] = acc[i][j][r];
            }
        }
}

template <bool XF, int WTR, int WTC>
__global__ __launch_bounds__(256) void pw_bwd_kernel(PwBwd p) {
    __shared__ __attribute__((aligned(16))) float lds[PW_LDS_FLOATS];
    const int b = blockIdx.x, nd = p.d_cx * p.d_ry;
    if (b < nd) {
        pw_dgrad_body<WTR, WTC>(lds, p, b % p.d_cx, b / p.d_cx);
    } else {
        const int wb = b - nd;
        const int bx = wb % p.w_cx, r = wb / p.w_cx;
        pw_wgrad_body<XF>(lds, p, bx, r % p.w_ny, r / p.w_ny);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Stem: downsample (Conv2d(1,3,3,padding=(1,3)) + bias, BatchNorm, ReLU, MaxPool2d((1,2))) and features[0]
// (Conv2d(3,32,3,stride 2,padding 1)), direct convolutions with one thread per pixel -- 3 and 27 inputs per pixel are far
// too thin for the tile engine, and an im2col matrix would be 9x the data.
//   z0: (B, H, W0, 3), W0 = T + 4;   pooled y0 = maxpool(relu(bn(z0))): (B, H, Wp, 3), Wp = W0 / 2, never stored;
//   z1: (B, H1, W1, 32).
// ---------------------------------------------------------------------------------------------------------
struct StemGeo {
    int B, H, T, W0, Wp, H1, W1;
};

// the nine inputs of downsample pixel (b, oh, ow): x[b, oh - 1 + kh, ow - 3 + kw], zero outside.  All nine loads are issued
// from clamped addresses BEFORE any of them is masked (two loops): written as one `inside ? x[..] : 0` per tap the compiler
// turns every tap into a branch around its load with a full wait behind it -- nine dependent memory round trips per pixel.
__device__ __forceinline__ void stem0_taps(const float* __restrict__ x, long sb, long sm, long st, const StemGeo& s, int b, int oh,
                                           int ow, float (&v)[9]) {
    const float* xb = x + b * sb;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const float* xr = xb + (long)min(max(oh - 1 + kh, 0), s.H - 1) * sm;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) v[kh * 3 + kw] = xr[(long)min(max(ow - 3 + kw, 0), s.T - 1) * st];
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int ih = oh - 1 + kh, it = ow - 3 + kw;
            if (ih < 0 || ih >= s.H || it < 0 || it >= s.T) v[kh * 3 + kw] = 0.0f;
        }
}

// z0 = conv(x) + bias, with the BatchNorm statistics of layer 0 (3 channels) reduced in the same launch
__global__ __launch_bounds__(256) void stem0_fwd_kernel(const float* __restrict__ x, long sb, long sm, long st,
                                                        const float* __restrict__ w, const float* __restrict__ bias, StemGeo s,
                                                        long pixels, int px_per_block, float* __restrict__ z0, Arrive arr,
                                                        FinFwd fin) {
    __shared__ float red[6][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float wk[3][9], bk[3];
#pragma unroll
    for (int co = 0; co < 3; ++co) {
        bk[co] = bias[co];
#pragma unroll
        for (int k = 0; k < 9; ++k) wk[co][k] = w[co * 9 + k];
    }
    float sum[3] = {0.f, 0.f, 0.f}, sq[3] = {0.f, 0.f, 0.f};
    const long p0 = (long)blockIdx.y * px_per_block;
    const long p1 = pixels < p0 + px_per_block ? pixels : p0 + px_per_block;
    for (long px = p0 + tid; px < p1; px += 256) {
        const int b = (int)((unsigned)px / (unsigned)(s.H * s.W0));     // pixel counts fit 31 bits (checked on the host)
        const int r = (int)px - b * s.H * s.W0;
        const int oh = r / s.W0, ow = r - oh * s.W0;
        float v[9];
        stem0_taps(x, sb, sm, st, s, b, oh, ow, v);
#pragma unroll
        for (int co = 0; co < 3; ++co) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 9; ++k) acc = fmaf(wk[co][k], v[k], acc);
            acc += bk[co];
            z0[px * 3 + co] = acc;
            sum[co] += acc;
            sq[co] = fmaf(acc, acc, sq[co]);
        }
    }
    if (arr.part1 == nullptr) return;
#pragma unroll
    for (int co = 0; co < 3; ++co) {
        const float a = wave_sum(sum[co]), q = wave_sum(sq[co]);
        if (lane == 0) {
            red[co][wave] = a;
            red[3 + co][wave] = q;
        }
    }
    __syncthreads();
    const bool own = tid < 3;
    const float v0 = own ? ((red[tid][0] + red[tid][1]) + red[tid][2]) + red[tid][3] : 0.0f;
    const float v1 = own ? ((red[3 + tid][0] + red[3 + tid][1]) + red[3 + tid][2]) + red[3 + tid][3] : 0.0f;
    if (publish_and_arrive(arr, 3, 0, blockIdx.y, gridDim.y, own, tid, v0, v1))
        finalize_fwd(arr.part2, arrive_rows(gridDim.y), 3, lane, fin);
}

// the 27 inputs of output pixel (b, oh, ow) of features[0]: pooled y0 at (2 oh - 1 + kh, 2 ow - 1 + kw), index ci*9 + kh*3 + kw
__device__ __forceinline__ void stem1_inputs(const float* __restrict__ z0, const float (&sc)[3], const float (&sh)[3],
                                             const StemGeo& s, int b, int oh, int ow, float (&v)[27]) {
    float t[9][6];     // all 54 loads first, from clamped addresses (see stem0_taps)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int ih = min(max(2 * oh - 1 + kh, 0), s.H - 1), iw = min(max(2 * ow - 1 + kw, 0), s.Wp - 1);
            const float* p = z0 + (((long)b * s.H + ih) * s.W0 + 2 * iw) * 3;
#pragma unroll
            for (int e = 0; e < 6; ++e) t[kh * 3 + kw][e] = p[e];
        }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int ih = 2 * oh - 1 + kh, iw = 2 * ow - 1 + kw;
            const bool inb = ih >= 0 && ih < s.H && iw >= 0 && iw < s.Wp;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const float a = fmaxf(fmaf(t[kh * 3 + kw][ci], sc[ci], sh[ci]), 0.0f);
                const float c = fmaxf(fmaf(t[kh * 3 + kw][3 + ci], sc[ci], sh[ci]), 0.0f);
                float y = fmaxf(a, c);
                if (!inb) y = 0.0f;
                v[ci * 9 + kh * 3 + kw] = y;
            }
        }
}

// z1 = conv(y0), stride 2, padding 1; y0 rebuilt from z0 on load.  One thread per output pixel, all 32 output channels; the
// weights are uniform across the wave (scalar operands).
__global__ __launch_bounds__(256) void stem1_fwd_kernel(const float* __restrict__ z0, const float* __restrict__ ss0,
                                                        const float* __restrict__ w, StemGeo s, long pixels,
                                                        float* __restrict__ z1) {
    const long px = (long)blockIdx.x * 256 + threadIdx.x;
    if (px >= pixels) return;
    float sc[3], sh[3];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) {
        sc[ci] = ss0[ci];
        sh[ci] = ss0[3 + ci];
    }
    const int b = (int)((unsigned)px / (unsigned)(s.H1 * s.W1));
    const int r = (int)px - b * s.H1 * s.W1;
    const int oh = r / s.W1, ow = r - oh * s.W1;
    float v[27];
    stem1_inputs(z0, sc, sh, s, b, oh, ow, v);
    float* out = z1 + px * 32;
#pragma unroll
    for (int c4 = 0; c4 < 8; ++c4) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int k = 0; k < 27; ++k) acc[e] = fmaf(w[(c4 * 4 + e) * 27 + k], v[k], acc[e]);
        *reinterpret_cast<float4*>(out + c4 * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

// Backward of features[0], ONE launch over the materialised dz1 (B, H1, W1, 32):
//   blocks [0, wblocks): weight gradient dW1[co][ci*9+tap] = sum_px dz1[px][co] * in[px][ci*9+tap].  A wave takes 64 pixels at a
//     time: every lane rebuilds its pixel's 27 inputs and loads its 32 dz, the wave parks them in LDS (two halves of 32
//     pixels) and multiplies in^T (32 x 32 px) by dz (32 px x 32) with the fp32 MFMA; slabs[block][co*27 + k].
//   the rest: data gradient to the pooled y0, un-pooled (first maximum wins, as PyTorch), through the ReLU, together with the
//     BatchNorm-backward reduction of layer 0: g0 (B, H, W0, 3) + partials.  Pooled pixels are taken by parity class
//     (ih % 2, iw % 2) -- a class shares its set of taps, so the weights stay scalar operands; a thread owns one pooled pixel =
//     two z0 columns.
struct Stem1Bwd {
    const float* dz1;
    const float* z0;
    const float* ss0;
    const float* w;        // (32, 27)
    StemGeo s;
    int wblocks, px_per_wblock;     // weight gradient
    int cls_blocks, px_per_thread;  // data gradient: blocks per parity class
    float* slabs;
    float* g0;
    Arrive arr;
    FinBwd fin;
};
constexpr int S1_LD = 72;   // floats per pixel row of the wave's LDS tile: 32 inputs | 32 dz | pad (72 = 8 mod 32)

__device__ __forceinline__ void stem1_wgrad_body(const Stem1Bwd& p, int wb) {
    __shared__ __attribute__((aligned(16))) float tile[4][32 * S1_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, l15 = lane & 15;
    const StemGeo& s = p.s;
    float sc[3], sh[3];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) {
        sc[ci] = p.ss0[ci];
        sh[ci] = p.ss0[3 + ci];
    }
    const long pixels = (long)s.B * s.H1 * s.W1;
    const long p0 = (long)wb * p.px_per_wblock;
    const long p1 = pixels < p0 + p.px_per_wblock ? pixels : p0 + p.px_per_wblock;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = {0.0f, 0.0f, 0.0f, 0.0f};
    float* my = tile[wave];
    for (long base = p0 + wave * 64; base < p1; base += 256) {
        const long px = base + lane;
        const bool ok = px < p1;
        const long pc = ok ? px : p1 - 1;
        const int b = (int)((unsigned)pc / (unsigned)(s.H1 * s.W1));
        const int r = (int)pc - b * s.H1 * s.W1;
        const int oh = r / s.W1, ow = r - oh * s.W1;
        float v[27];
        stem1_inputs(p.z0, sc, sh, s, b, oh, ow, v);
        float4 d[8];
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4) d[c4] = ldg4(p.dz1 + pc * 32 + c4 * 4);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if ((lane >> 5) == half) {
                float* row = my + (lane & 31) * S1_LD;
#pragma unroll
                for (int k = 0; k < 27; ++k) row[k] = ok ? v[k] : 0.0f;
#pragma unroll
                for (int k = 27; k < 32; ++k) row[k] = 0.0f;
#pragma unroll
                for (int c4 = 0; c4 < 8; ++c4) *reinterpret_cast<float4*>(row + 32 + c4 * 4) = ok ? d[c4] : f4_zero();
            }
            wave_lds_sync();
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const float* row = my + (4 * ks + q) * S1_LD;
                float a[2], bb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = row[16 * i + l15];
#pragma unroll
                for (int j = 0; j < 2; ++j) bb[j] = row[32 + 16 * j + l15];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bb[j], acc[i][j], 0, 0, 0);
            }
            wave_lds_sync();
        }
    }
    // the four waves' 32 x 32 partial products -> one slab row (fixed order)
    __syncthreads();
    float* red = &tile[0][0];     // [wave][k (32)][co (32)]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wave * 32 + 16 * i + 4 * q + r) * 32 + 16 * j + l15] = acc[i][j][r];
    __syncthreads();
    for (int o = tid; o < 32 * 27; o += 256) {
        const int co = o / 27, k = o - co * 27;
        const float t = ((red[(0 * 32 + k) * 32 + co] + red[(1 * 32 + k) * 32 + co]) + red[(2 * 32 + k) * 32 + co]) +
                        red[(3 * 32 + k) * 32 + co];
        p.slabs[(size_t)wb * (32 * 27) + o] = t;
    }
}

__device__ __forceinline__ void stem1_dgrad_body(const Stem1Bwd& p, int db) {
    __shared__ float red[6][4];
    __shared__ float wl[32 * 27];     // the weights: every thread of the block reads the same taps (LDS broadcast)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const StemGeo& s = p.s;
    const int cls = db / p.cls_blocks, cb = db - cls * p.cls_blocks;
    const int ph = cls >> 1, pw = cls & 1;
    const int na = (s.H + 1) / 2, nc = (s.Wp + 1) / 2;
    const long per_class = (long)s.B * na * nc;
    float sc[3], sh[3], me[3], rs[3];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) {
        sc[ci] = p.ss0[ci];
        sh[ci] = p.ss0[3 + ci];
        me[ci] = p.ss0[6 + ci];
        rs[ci] = p.ss0[9 + ci];
    }
    for (int i = tid; i < 32 * 27; i += 256) wl[i] = p.w[i];
    __syncthreads();
    float s1[3] = {0.f, 0.f, 0.f}, s2[3] = {0.f, 0.f, 0.f};
    for (int it = 0; it < p.px_per_thread; ++it) {
        const long idx = ((long)cb * p.px_per_thread + it) * 256 + tid;
        if (idx >= per_class) continue;      // (no barrier inside the loop)
        const int b = (int)((unsigned)idx / (unsigned)(na * nc));
        const int r = (int)idx - b * na * nc;
        const int a = r / nc, c = r - a * nc;
        const int ih = 2 * a + ph, iw = 2 * c + pw;
        if (ih >= s.H || iw >= s.Wp) continue;
        // gradient of the pooled pixel: outputs (oh, ow) with 2 oh - 1 + kh = ih, 2 ow - 1 + kw = iw
        float dy[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            if (hh == 1 && ph == 0) break;
            const int kh = ph == 0 ? 1 : (hh == 0 ? 0 : 2);
            const int oh = ph == 0 ? a : (hh == 0 ? a + 1 : a);
#pragma unroll
            for (int ww = 0; ww < 2; ++ww) {
                if (ww == 1 && pw == 0) break;
                const int kw = pw == 0 ? 1 : (ww == 0 ? 0 : 2);
                const int ow = pw == 0 ? c : (ww == 0 ? c + 1 : c);
                const bool ok = oh < s.H1 && ow < s.W1;
                const float* dp = p.dz1 + (((long)b * s.H1 + min(oh, s.H1 - 1)) * s.W1 + min(ow, s.W1 - 1)) * 32;
                float4 d[8];
#pragma unroll
                for (int c4 = 0; c4 < 8; ++c4) d[c4] = ldg4(dp + c4 * 4);
                const int tap = kh * 3 + kw;     // uniform over the block (parity class)
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    float t = 0.0f;
#pragma unroll
                    for (int c4 = 0; c4 < 8; ++c4)
#pragma unroll
                        for (int e = 0; e < 4; ++e) t = fmaf(f4_get(d[c4], e), wl[(c4 * 4 + e) * 27 + ci * 9 + tap], t);
                    dy[ci] += ok ? t : 0.0f;
                }
            }
        }
        // through the max pool and the ReLU to the two z0 columns of this pooled pixel
        const long zoff = (((long)b * s.H + ih) * s.W0 + 2 * iw) * 3;
        float t[6], gq[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) t[e] = p.z0[zoff + e];
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
            const float ya = fmaf(t[ci], sc[ci], sh[ci]), yb = fmaf(t[3 + ci], sc[ci], sh[ci]);
            const bool first = fmaxf(ya, 0.0f) >= fmaxf(yb, 0.0f);     // ties: the earlier index wins
            gq[ci] = (first && ya > 0.0f) ? dy[ci] : 0.0f;
            gq[3 + ci] = (!first && yb > 0.0f) ? dy[ci] : 0.0f;
            s1[ci] += gq[ci] + gq[3 + ci];
            s2[ci] = fmaf(gq[ci], (t[ci] - me[ci]) * rs[ci], s2[ci]);
            s2[ci] = fmaf(gq[3 + ci], (t[3 + ci] - me[ci]) * rs[ci], s2[ci]);
        }
#pragma unroll
        for (int e = 0; e < 6; ++e) p.g0[zoff + e] = gq[e];
        if (iw == s.Wp - 1) {     // an odd last column of z0 has no pooled pixel: no gradient
            for (int col = 2 * s.Wp; col < s.W0; ++col)
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) p.g0[(((long)b * s.H + ih) * s.W0 + col) * 3 + ci] = 0.0f;
        }
    }
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) {
        const float a = wave_sum(s1[ci]), qq = wave_sum(s2[ci]);
        if (lane == 0) {
            red[ci][wave] = a;
            red[3 + ci][wave] = qq;
        }
    }
    __syncthreads();
    const bool own = tid < 3;
    const float v0 = own ? ((red[tid][0] + red[tid][1]) + red[tid][2]) + red[tid][3] : 0.0f;
    const float v1 = own ? ((red[3 + tid][0] + red[3 + tid][1]) + red[3 + tid][2]) + red[3 + tid][3] : 0.0f;
    const int R1 = 4 * p.cls_blocks;
    if (publish_and_arrive(p.arr, 3, 0, db, R1, own, tid, v0, v1)) finalize_bwd(p.arr.part2, arrive_rows(R1), 3, lane, p.fin);
}

__global__ __launch_bounds__(256) void stem1_bwd_kernel(Stem1Bwd p) {
    if ((int)blockIdx.x < p.wblocks)
        stem1_wgrad_body(p, blockIdx.x);
    else
        stem1_dgrad_body(p, blockIdx.x - p.wblocks);
}

// Backward of the downsample convolution: dz0 from (g0, z0, bc0) per pixel; dW0[co][tap] = sum dz0[co] * x[tap], db0[co] = sum
// dz0[co].  slab_w[block][27], slab_b[block][3].
__global__ __launch_bounds__(256) void stem0_bwd_kernel(const float* __restrict__ x, long sb, long sm, long st,
                                                        const float* __restrict__ g0, const float* __restrict__ z0,
                                                        const float* __restrict__ bc0, StemGeo s, long pixels, int px_per_block,
                                                        float* __restrict__ slab_w, float* __restrict__ slab_b) {
    __shared__ float red[30][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float ksc[3], kc1[3], kc0[3];
#pragma unroll
    for (int co = 0; co < 3; ++co) {
        ksc[co] = bc0[co];
        kc1[co] = bc0[3 + co];
        kc0[co] = bc0[6 + co];
    }
    float acc[30];
#pragma unroll
    for (int k = 0; k < 30; ++k) acc[k] = 0.0f;
    const long p0 = (long)blockIdx.x * px_per_block;
    const long p1 = pixels < p0 + px_per_block ? pixels : p0 + px_per_block;
    for (long px = p0 + tid; px < p1; px += 256) {
        const int b = (int)((unsigned)px / (unsigned)(s.H * s.W0));     // pixel counts fit 31 bits (checked on the host)
        const int r = (int)px - b * s.H * s.W0;
        const int oh = r / s.W0, ow = r - oh * s.W0;
        float v[9], dz[3];
        stem0_taps(x, sb, sm, st, s, b, oh, ow, v);
#pragma unroll
        for (int co = 0; co < 3; ++co) dz[co] = fmaf(ksc[co], g0[px * 3 + co], fmaf(kc1[co], z0[px * 3 + co], kc0[co]));
#pragma unroll
        for (int co = 0; co < 3; ++co) {
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[co * 9 + k] = fmaf(dz[co], v[k], acc[co * 9 + k]);
            acc[27 + co] += dz[co];
        }
    }
#pragma unroll
    for (int k = 0; k < 30; ++k) {
        const float t = wave_sum(acc[k]);
        if (lane == 0) red[k][wave] = t;
    }
    __syncthreads();
    if (tid < 30) {
        const float t = ((red[tid][0] + red[tid][1]) + red[tid][2]) + red[tid][3];
        if (tid < 27)
            slab_w[(size_t)blockIdx.x * 27 + tid] = t;
        else
            slab_b[(size_t)blockIdx.x * 3 + (tid - 27)] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Depthwise 3x3 (padding 1).  All three kernels share one geometry: block = 64 channels (lane = channel: every load of a
// wave is 256 contiguous bytes) x one chunk of IMAGE rows; the 4 waves take the rows of the chunk in turn and walk a row
// in segments of DW_SEG = 4 outputs that share one window of inputs (3 rows x 6 columns at stride 1, 3 x 9 at stride 2),
// loaded unconditionally from clamped addresses and masked afterwards.  A lane's channel never changes, so the nine
// weights and every BatchNorm constant it needs are registers, and per-channel sums are running registers that leave the
// block as one row of partials.
// ---------------------------------------------------------------------------------------------------------
// forward: z[b,oh,ow,c] = sum_tap w[c*9+tap] * y_in[b, oh*s-1+kh, ow*s-1+kw, c],  y_in = relu6(z_in * scale + shift) inside
// the image, 0 in the padding; taps accumulate in the order (kh, kw) with fmaf.
template <int STRIDE>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const float* __restrict__ x, const float* __restrict__ ss_in,
                                                     const float* __restrict__ w, int H, int W, int C, int Ho, int Wo, int nrows,
                                                     int rows_per_chunk, float* __restrict__ z, Arrive arr, FinFwd fin) {
    constexpr int NIN = (DW_SEG - 1) * STRIDE + 3;     // input columns under DW_SEG outputs
    __shared__ float red[2][4][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    // layers of <= 32 channels (the 266 k-pixel first block) put TWO image rows into a wave, 32 lanes each
    const int rsub = C <= 32 ? 2 : 1, rs = rsub == 2 ? lane >> 5 : 0;
    const int c = blockIdx.x * 64 + (rsub == 2 ? lane & 31 : lane);
    const bool cok = c < C;
    const int cc = cok ? c : C - 1;
    float wk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k] = w[cc * 9 + k];
    const float sc = ss_in[cc], sh = ss_in[C + cc];
    const int r0 = blockIdx.y * rows_per_chunk;
    const int r1 = min(nrows, r0 + rows_per_chunk);
    float s = 0.0f, q = 0.0f;
    for (int t = r0 + rg * rsub + rs; t < r1; t += 4 * rsub) {
        const int b = t / Ho, oh = t - b * Ho;
        const float* xb = x + (long)b * H * W * C + cc;
        float* zr = z + (long)t * Wo * C + cc;
        // A segment of DW_SEG outputs.  FULL segments (all DW_SEG columns inside the row) store unconditionally -- lanes past the
        // last channel are clamped to it and write its values again -- so that the loop body has no branch around a store: with one,
        // the compiler waits for vmcnt(0) before the next segment's arithmetic, i.e. for the stores just issued (vmcnt counts
        // loads and stores alike), and the store and load streams of a wave take turns instead of overlapping.
        auto segment = [&](int ow0, auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            float v[3][NIN];
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
            for (int kh = 0; kh < 3; ++kh) {
                const int ih = oh * STRIDE - 1 + kh;
#pragma unroll
                for (int j = 0; j < NIN; ++j) {
                    const int iw = ow0 * STRIDE - 1 + j;
                    const bool inb = ih >= 0 && ih < H && iw >= 0 && iw < W;
                    v[kh][j] = inb ? relu6f(fmaf(v[kh][j], sc, sh)) : 0.0f;
                }
            }
#pragma unroll
            for (int o = 0; o < DW_SEG; ++o) {
                float acc = 0.0f;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) acc = fmaf(wk[kh * 3 + kw], v[kh][o * STRIDE + kw], acc);
                if (FULL) {
                    zr[(long)(ow0 + o) * C] = acc;
                    s += acc;
                    q = fmaf(acc, acc, q);
                } else if (ow0 + o < Wo) {
                    if (cok) zr[(long)(ow0 + o) * C] = acc;
                    s += acc;
                    q = fmaf(acc, acc, q);
                }
            }
        };
        int ow0 = 0;
        for (; ow0 + DW_SEG <= Wo; ow0 += DW_SEG) segment(ow0, std::true_type{});
        if (ow0 < Wo) segment(ow0, std::false_type{});
    }
    if (arr.part1 == nullptr) return;
    if (rsub == 2) {
        s += __shfl_xor(s, 32);
        q += __shfl_xor(q, 32);
    }
    red[0][rg][lane] = s;
    red[1][rg][lane] = q;
    __syncthreads();
    const bool own = rg == 0 && rs == 0 && cok;
    const float v0 = ((red[0][0][lane] + red[0][1][lane]) + red[0][2][lane]) + red[0][3][lane];
    const float v1 = ((red[1][0][lane] + red[1][1][lane]) + red[1][2][lane]) + red[1][3][lane];
    if (publish_and_arrive(arr, C, blockIdx.x, blockIdx.y, gridDim.y, own, c, v0, v1))
        finalize_fwd(arr.part2, arrive_rows(gridDim.y), C, blockIdx.x * 64 + lane, fin);     // one lane per column
}

// data gradient over the INPUT rows (b, ih): dy_j[b,ih,iw,c] = sum_tap w[c*9+tap] * dz[b,oh,ow,c] over the outputs whose
// window holds (ih, iw) at that tap (oh*s - 1 + kh = ih, ow*s - 1 + kw = iw); dz rebuilt from (g_k, z_k, bc_k) per loaded
// window element; the gradient window under 4 inputs is 3 rows x 6 columns at stride 1, 2 x 4 at stride 2 (loaded for every
// tap parity, masked).  Epilogue = EpiBwd for the layer in front (same channels).
struct DwBwd {
    const float* g;        // g_k, z_k: (B, Ho, Wo, C); bc_k: 3 C
    const float* zk;
    const float* bc;
    const float* w;        // (C, 9)
    const float* x;        // z_{k-1} (B, H, W, C) with ss_in
    const float* ss_in;
    int H, W, C, Ho, Wo;
    int ncb;                          // channel blocks
    int d_rows, d_rpc, d_chunks;      // data gradient: image rows of the INPUT, rows per chunk, chunks
    int w_rows, w_rpc, w_chunks;      // weight gradient: image rows of the OUTPUT
    EpiBwd e;
    Arrive arr;
    FinBwd fin;
    float* slabs;
};

template <int STRIDE>
__device__ __forceinline__ void dw_dgrad_body(const DwBwd& p, int cb, int by) {
    const float* __restrict__ g = p.g;
    const float* __restrict__ zk = p.zk;
    const float* __restrict__ bc = p.bc;
    const float* __restrict__ w = p.w;
    const int H = p.H, W = p.W, C = p.C, Ho = p.Ho, Wo = p.Wo, nrows = p.d_rows, rows_per_chunk = p.d_rpc;
    const EpiBwd& e = p.e;
    constexpr int NCOL = STRIDE == 1 ? DW_SEG + 2 : (DW_SEG + 1) / STRIDE + 2;   // 6 at stride 1, 4 at stride 2
    __shared__ float red[2][4][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    // layers of <= 32 channels (the 266 k-pixel first block) put TWO image rows into a wave, 32 lanes each
    const int rsub = C <= 32 ? 2 : 1, rs = rsub == 2 ? lane >> 5 : 0;
    const int c = cb * 64 + (rsub == 2 ? lane & 31 : lane);
    const bool cok = c < C;
    const int cc = cok ? c : C - 1;
    float wk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k] = w[cc * 9 + k];
    const float ksc = bc[cc], kc1 = bc[C + cc], kc0 = bc[2 * C + cc];
    const float jsc = e.ssj[cc], jsh = e.ssj[C + cc], jme = e.ssj[2 * C + cc], jrs = e.ssj[3 * C + cc];
    const int r0 = by * rows_per_chunk;
    const int r1 = min(nrows, r0 + rows_per_chunk);
    float s1 = 0.0f, s2 = 0.0f;
    for (int t = r0 + rg * rsub + rs; t < r1; t += 4 * rsub) {
        const int b = t / H, ih = t - b * H;
        const float* gb = g + (long)b * Ho * Wo * C + cc;
        const float* zb = zk + (long)b * Ho * Wo * C + cc;
        const float* zjr = e.zj + (long)t * W * C + cc;
        float* gjr = e.gj + (long)t * W * C + cc;
        bool okh[3];
        int ohs[3];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int nh = ih + 1 - kh;                                // = oh * s
            const int oh = (nh + STRIDE) / STRIDE - 1;                 // floor division for nh >= -s
            okh[kh] = nh >= 0 && oh * STRIDE == nh && oh < Ho;
            ohs[kh] = min(max(oh, 0), Ho - 1);
        }
        // (FULL segments store unconditionally, clamped lanes repeating the last channel's values: see dw_fwd_kernel)
        auto segment = [&](int iw0, auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            // gradient columns that can touch inputs iw0 .. iw0+3: ow = (iw + 1 - kw) / s -> from floor((iw0 - 1) / s) upwards
            const int owb = (iw0 - 1 + STRIDE) / STRIDE - 1;
            float v[3][NCOL], vz[3][NCOL], zj[DW_SEG];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int j = 0; j < NCOL; ++j) {
                    const int owc = min(max(owb + j, 0), Wo - 1);
                    v[kh][j] = gb[((long)ohs[kh] * Wo + owc) * C];
                    vz[kh][j] = zb[((long)ohs[kh] * Wo + owc) * C];
                }
#pragma unroll
            for (int o = 0; o < DW_SEG; ++o) zj[o] = zjr[(long)min(iw0 + o, W - 1) * C];
            // dz of the window, zero outside the gradient map (rows by okh, columns by their slot)
            bool okc[NCOL];
#pragma unroll
            for (int j = 0; j < NCOL; ++j) okc[j] = owb + j >= 0 && owb + j < Wo;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int j = 0; j < NCOL; ++j)
                    v[kh][j] = (okh[kh] && okc[j]) ? fmaf(ksc, v[kh][j], fmaf(kc1, vz[kh][j], kc0)) : 0.0f;
            // Which window slot a tap of input column iw0 + o reads is a compile-time fact (iw0 is a multiple of DW_SEG = 4, so the
            // parity of iw0 + o + 1 - kw is that of o + 1 - kw): slot o + 2 - kw at stride 1; at stride 2 only taps with an even
            // d = o + 1 - kw have an output column, slot d / 2 + 1.  (Measured: 3 % on the launch, which is not VALU-bound -- nor
            // bound by address arithmetic: scalar row addresses changed nothing; tools/mb_variants.py: the data-gradient blocks are
            // 90 % of the launch.)
#pragma unroll
            for (int o = 0; o < DW_SEG; ++o) {
                const int iw = iw0 + o;
                float acc = 0.0f;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        if (STRIDE == 1) {
                            acc = fmaf(wk[kh * 3 + kw], v[kh][o + 2 - kw], acc);
                        } else {
                            const int d = o + 1 - kw;
                            if (d >= 0 && (d & 1) == 0) acc = fmaf(wk[kh * 3 + kw], v[kh][d / 2 + 1], acc);
                        }
                    }
                if (FULL || iw < W) {
                    const bool pass = mb_act_passes(fmaf(zj[o], jsc, jsh), e.act);
                    const float gg = pass ? acc : 0.0f;
                    if (FULL || cok) gjr[(long)iw * C] = gg;
                    s1 += gg;
                    s2 = fmaf(gg, (zj[o] - jme) * jrs, s2);
                }
            }
        };
        int iw0 = 0;
        for (; iw0 + DW_SEG <= W; iw0 += DW_SEG) segment(iw0, std::true_type{});
        if (iw0 < W) segment(iw0, std::false_type{});
    }
    if (rsub == 2) {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
    }
    red[0][rg][lane] = s1;
    red[1][rg][lane] = s2;
    __syncthreads();
    const bool own = rg == 0 && rs == 0 && cok;
    const float v0 = ((red[0][0][lane] + red[0][1][lane]) + red[0][2][lane]) + red[0][3][lane];
    const float v1 = ((red[1][0][lane] + red[1][1][lane]) + red[1][2][lane]) + red[1][3][lane];
    if (publish_and_arrive(p.arr, C, cb, by, p.d_chunks, own, c, v0, v1))
        finalize_bwd(p.arr.part2, arrive_rows(p.d_chunks), C, cb * 64 + lane, p.fin);     // one lane per column
}

// weight gradient: dW[c][tap] = sum_{b,oh,ow} dz[.,c] * y_in[shifted, c]; both factors rebuilt on load.
// part[chunk][c*9+tap]; the chunks are folded in a fixed order by the deferred slab sum.
template <int STRIDE>
__device__ __forceinline__ void dw_wgrad_body(const DwBwd& p, int cb, int by) {
    const float* __restrict__ g = p.g;
    const float* __restrict__ zk = p.zk;
    const float* __restrict__ bc = p.bc;
    const float* __restrict__ x = p.x;
    const float* __restrict__ ss_in = p.ss_in;
    float* __restrict__ part = p.slabs;
    const int H = p.H, W = p.W, C = p.C, Ho = p.Ho, Wo = p.Wo, nrows = p.w_rows, rows_per_chunk = p.w_rpc;
    constexpr int NIN = (DW_SEG - 1) * STRIDE + 3;
    __shared__ float red[4][9][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    // layers of <= 32 channels (the 266 k-pixel first block) put TWO image rows into a wave, 32 lanes each
    const int rsub = C <= 32 ? 2 : 1, rs = rsub == 2 ? lane >> 5 : 0;
    const int c = cb * 64 + (rsub == 2 ? lane & 31 : lane);
    const bool cok = c < C;
    const int cc = cok ? c : C - 1;
    const float ksc = bc[cc], kc1 = bc[C + cc], kc0 = bc[2 * C + cc];
    const float sc = ss_in[cc], sh = ss_in[C + cc];
    const int r0 = by * rows_per_chunk;
    const int r1 = min(nrows, r0 + rows_per_chunk);
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.0f;
    for (int t = r0 + rg * rsub + rs; t < r1; t += 4 * rsub) {
        const int b = t / Ho, oh = t - b * Ho;
        const float* xb = x + (long)b * H * W * C + cc;
        const float* gr = g + (long)t * Wo * C + cc;
        const float* zr = zk + (long)t * Wo * C + cc;
        for (int ow0 = 0; ow0 < Wo; ow0 += DW_SEG) {
            float d[DW_SEG], dzv[DW_SEG], v[3][NIN];
#pragma unroll
            for (int o = 0; o < DW_SEG; ++o) {
                d[o] = gr[(long)min(ow0 + o, Wo - 1) * C];
                dzv[o] = zr[(long)min(ow0 + o, Wo - 1) * C];
            }
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
            for (int o = 0; o < DW_SEG; ++o) d[o] = ow0 + o < Wo ? fmaf(ksc, d[o], fmaf(kc1, dzv[o], kc0)) : 0.0f;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int ih = oh * STRIDE - 1 + kh;
#pragma unroll
                for (int j = 0; j < NIN; ++j) {
                    const int iw = ow0 * STRIDE - 1 + j;
                    const bool inb = ih >= 0 && ih < H && iw >= 0 && iw < W;
                    v[kh][j] = inb ? relu6f(fmaf(v[kh][j], sc, sh)) : 0.0f;
                }
            }
#pragma unroll
            for (int o = 0; o < DW_SEG; ++o)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) acc[kh * 3 + kw] = fmaf(d[o], v[kh][o * STRIDE + kw], acc[kh * 3 + kw]);
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) red[rg][t][lane] = rsub == 2 ? acc[t] + __shfl_xor(acc[t], 32) : acc[t];
    __syncthreads();
    if (rg == 0 && rs == 0 && cok) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
            part[(size_t)by * C * 9 + c * 9 + t] = ((red[0][t][lane] + red[1][t][lane]) + red[2][t][lane]) + red[3][t][lane];
    }
}

// Backward of a depthwise layer, ONE launch: blocks [0, ncb * d_chunks) are the data gradient (+ BatchNorm-backward reduction
// of the layer in front), the rest the weight gradient.
template <int STRIDE>
__global__ __launch_bounds__(256) void dw_bwd_kernel(DwBwd p) {
    const int b = blockIdx.x, nd = p.ncb * p.d_chunks;
    if (b < nd) {
        dw_dgrad_body<STRIDE>(p, b % p.ncb, b / p.ncb);
    } else {
        const int wb = b - nd;
        dw_wgrad_body<STRIDE>(p, wb % p.ncb, wb / p.ncb);
    }
}

// ---------------------------------------------------------------------------------------------------------
// network tail: adaptive_avg_pool2d(1) over relu6(bn(z_last)) (+ dropout), and its backward fused with the last layer's
// BatchNorm-backward reduction
// ---------------------------------------------------------------------------------------------------------
// One block per utterance: pooled[b][c] = mean_p relu6(bn(z_last[b,p,c])), pooled_d = dropout(pooled), and the classifier
// logits[b][l] = pooled_d[b] . W[l] + bias[l] (a (B x 1280) x (1280 x L) product with L ~ 12: as a tile GEMM it was 80 dependent
// k-steps on 8 blocks).  A thread keeps its channels' pooled_d in registers; per label: partial dot product, wave sum, four
// wave results folded in order.
constexpr int POOL_CPT = 8;     // channels per thread: up to 2048 channels per block
__global__ __launch_bounds__(256) void pool_classify_kernel(const float* __restrict__ z, const float* __restrict__ ss, int HW, int C,
                                                            const float* __restrict__ mask, float scale,
                                                            const float* __restrict__ wc, const float* __restrict__ bias, int L,
                                                            float* __restrict__ pooled, float* __restrict__ pooled_d,
                                                            float* __restrict__ logits) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long b = blockIdx.x;
    float pd[POOL_CPT];
#pragma unroll
    for (int u = 0; u < POOL_CPT; ++u) {
        const int c = tid + 256 * u;
        pd[u] = 0.0f;
        if (c < C) {
            const float sc = ss[c], sh = ss[C + c];
            float acc = 0.0f;
            for (int p = 0; p < HW; ++p) acc += relu6f(fmaf(z[(b * HW + p) * C + c], sc, sh));
            const float v = acc / (float)HW;
            pooled[b * C + c] = v;
            pd[u] = mask != nullptr ? v * mask[b * C + c] * scale : v;
            pooled_d[b * C + c] = pd[u];
        }
    }
    for (int l = 0; l < L; ++l) {
        float t = 0.0f;
#pragma unroll
        for (int u = 0; u < POOL_CPT; ++u) {
            const int c = tid + 256 * u;
            t = fmaf(pd[u], c < C ? wc[(long)l * C + c] : 0.0f, t);
        }
        t = wave_sum(t);
        __syncthreads();     // the previous label's readers are done with `red`
        if (lane == 0) red[wave] = t;
        __syncthreads();
        if (tid == 0) logits[b * L + l] = (((red[0] + red[1]) + red[2]) + red[3]) + bias[l];
    }
}

// dpooled[b,c] = sum_l dlogits[b,l] * W[l,c] (the classifier's data gradient, L ~ 12 terms, rebuilt per element);
// dy[b,p,c] = dpooled[b,c] (* mask * scale) / HW -> g, partial sums, finalize (block = 64 channels x a chunk of pixel rows)
__global__ __launch_bounds__(256) void avgpool_bwd_reduce_kernel(const float* __restrict__ dlogits, const float* __restrict__ wc,
                                                                 int L, const float* __restrict__ mask, float scale, int HW, int C,
                                                                 int B, int utts_per_chunk, EpiBwd e, Arrive arr, FinBwd fin) {
    __shared__ float red[2][4][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool cok = c < C;
    const int cc = cok ? c : C - 1;
    const float jsc = e.ssj[cc], jsh = e.ssj[C + cc], jme = e.ssj[2 * C + cc], jrs = e.ssj[3 * C + cc];
    const int b0 = blockIdx.y * utts_per_chunk;
    const int b1 = min(B, b0 + utts_per_chunk);
    float s1 = 0.0f, s2 = 0.0f;
    for (long b = b0 + rg; b < b1; b += 4) {     // a wave takes an utterance: its gradient once, then its HW pixels
        float d = 0.0f;
        for (int l = 0; l < L; ++l) d = fmaf(dlogits[b * L + l], wc[(long)l * C + cc], d);
        if (mask != nullptr) d *= mask[b * C + cc] * scale;
        d = d / (float)HW;
        for (int px = 0; px < HW; ++px) {
            const long m = b * HW + px;
            const float zv = e.zj[m * C + cc];
            const float gg = mb_act_passes(fmaf(zv, jsc, jsh), e.act) ? d : 0.0f;
            if (cok) e.gj[m * C + c] = gg;
            s1 += gg;
            s2 = fmaf(gg, (zv - jme) * jrs, s2);
        }
    }
    red[0][rg][lane] = s1;
    red[1][rg][lane] = s2;
    __syncthreads();
    const bool own = rg == 0 && cok;
    const float v0 = ((red[0][0][lane] + red[0][1][lane]) + red[0][2][lane]) + red[0][3][lane];
    const float v1 = ((red[1][0][lane] + red[1][1][lane]) + red[1][2][lane]) + red[1][3][lane];
    if (publish_and_arrive(arr, C, blockIdx.x, blockIdx.y, gridDim.y, own, c, v0, v1))
        finalize_bwd(arr.part2, arrive_rows(gridDim.y), C, c, fin);
}

// ---------------------------------------------------------------------------------------------------------
// elementwise pieces (materialised outputs, stem)
// ---------------------------------------------------------------------------------------------------------
// dz = scale * g + c1 * z + c0 in memory (features[0] only: both roles of its backward launch read dz many times)
__global__ void dz_apply_kernel(const float* __restrict__ g, const float* __restrict__ z, const float* __restrict__ bc, int C,
                                long total, float* __restrict__ dz) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = chan_of(idx, C);
        dz[idx] = fmaf(bc[c], g[idx], fmaf(bc[C + c], z[idx], bc[2 * C + c]));
    }
}

// eval mode: ss of every layer from the running statistics, one launch (blockIdx.y = layer)
struct EvalJobs {
    int c[64];
    long long gamma_off[64], rmean_off[64], ss_off[64];
};
__global__ void bn_eval_ss_kernel(EvalJobs jobs, const float* __restrict__ params, const float* __restrict__ buffers,
                                  float* __restrict__ ws) {
    const int k = blockIdx.y;
    const int C = jobs.c[k];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float* gamma = params + jobs.gamma_off[k];     // beta follows gamma, running_var follows running_mean
    const float* rmean = buffers + jobs.rmean_off[k];
    float* ss = ws + jobs.ss_off[k];
    const float mean = rmean[c];
    const float rstd = 1.0f / sqrtf(rmean[C + c] + MB_EPS);
    const float scale = gamma[c] * rstd;
    ss[c] = scale;
    ss[C + c] = gamma[C + c] - mean * scale;
    ss[2 * C + c] = mean;
    ss[3 * C + c] = rstd;
}

// BatchNorm statistics of a narrow (M x C) matrix, C <= 32 (features[0]'s 32 channels: its kernel has one thread per pixel
// and all 32 channels in registers -- reducing 64 sums across the block there would cost more than this pass).
// Lane -> (row slot rs = lane / cp, column lane % cp), cp = col_pack(C); rows advance by 4 waves x (64/cp) slots, four loads
// in flight; (sum z, sum z^2) -> partial row -> finalize_fwd by the last block.
__global__ __launch_bounds__(256) void col_stats_kernel(const float* __restrict__ z, long rows, int C, int cp, long rows_per_chunk,
                                                        Arrive arr, FinFwd fin) {
    __shared__ float red[2][4][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int rsub = 64 / cp;
    const int c = lane & (cp - 1), rs = lane / cp;
    const long r0 = (long)blockIdx.y * rows_per_chunk;
    const long r1 = rows < r0 + rows_per_chunk ? rows : r0 + rows_per_chunk;
    float s0 = 0.0f, s1 = 0.0f;
    if (c < C) {
        const long step = 4L * rsub;
        double t0 = 0.0, t1 = 0.0;
        float a0[4], a1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a0[u] = a1[u] = 0.0f;
        int since_flush = 0;
        for (long m = r0 + (long)rg * rsub + rs; m < r1; m += 4 * step) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long mm = m + u * step;
                const float t = z[(mm < r1 ? mm : m) * C + c];
                v[u] = mm < r1 ? t : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0[u] += v[u];
                a1[u] = fmaf(v[u], v[u], a1[u]);
            }
            if (++since_flush == 16) {   // short fp32 runs, fp64 totals
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    t0 += (double)a0[u];
                    t1 += (double)a1[u];
                    a0[u] = a1[u] = 0.0f;
                }
                since_flush = 0;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            t0 += (double)a0[u];
            t1 += (double)a1[u];
        }
        s0 = (float)t0;
        s1 = (float)t1;
    }
    red[0][rg][lane] = s0;
    red[1][rg][lane] = s1;
    __syncthreads();
    const bool own = rg == 0 && rs == 0 && c < C;
    float v0 = 0.0f, v1 = 0.0f;
    if (own) {
        double t0 = 0.0, t1 = 0.0;
        for (int w = 0; w < 4; ++w)
            for (int qq = 0; qq < rsub; ++qq) {
                t0 += (double)red[0][w][qq * cp + c];
                t1 += (double)red[1][w][qq * cp + c];
            }
        v0 = (float)t0;
        v1 = (float)t1;
    }
    if (publish_and_arrive(arr, C, 0, blockIdx.y, gridDim.y, own, c, v0, v1)) finalize_fwd(arr.part2, arrive_rows(gridDim.y), C, lane, fin);
}

// ---------------------------------------------------------------------------------------------------------
// Deferred slab sums: every weight gradient of a backward call leaves `nparts` slabs of n floats; ONE launch folds them all
// (a block owns 64 consecutive outputs of one job, found through the jobs' block prefix; its 4 waves take the slabs
// wave, wave+4, ..., four loads in flight, and combine through LDS in a fixed order).
// ---------------------------------------------------------------------------------------------------------
struct MbSumJob {
    const float* part;
    float* out;
    int n, nparts, blk0, pad;
};
struct MbSumJobs {
    MbSumJob j[MB_MAX_JOBS];
    int count, blocks;
};
__global__ __launch_bounds__(256) void sum_jobs_kernel(MbSumJobs jobs) {
    __shared__ float red[4][64];
    int lo = 0, hi = jobs.count - 1;
    while (lo < hi) {   // last job whose first block is <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (jobs.j[mid].blk0 <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const MbSumJob& jb = jobs.j[lo];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const long n = jb.n;
    const long i = (long)((int)blockIdx.x - jb.blk0) * 64 + lane;
    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (i < n) {
        // eight slabs per trip, requested together from clamped rows and masked afterwards: jobs with few slabs (the wide late
        // layers have 7-24) are ONE round trip per wave instead of a chain of dependent ones
        int gq = rg;
        // the first blocks' layers have ~1000 slabs of a few hundred outputs: sixteen in flight per lane (their fold is a chain of
        // dependent round trips: 33 of them at eight per lane were most of this launch's 58 us)
        for (; gq + 64 <= jb.nparts; gq += 64) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = jb.part[(long)(gq + 4 * u) * n + i];
#pragma unroll
            for (int u = 0; u < 16; ++u) s[u & 3] += v[u];
        }
        for (; gq < jb.nparts; gq += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int g = gq + 4 * u < jb.nparts ? gq + 4 * u : jb.nparts - 1;
                v[u] = jb.part[(long)g * n + i];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u & 3] += gq + 4 * u < jb.nparts ? v[u] : 0.0f;
        }
    }
    red[rg][lane] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (rg == 0 && i < n) jb.out[i] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}
struct JobList {
    MbSumJobs jobs;
    JobList() { jobs.count = 0; jobs.blocks = 0; }
    void add(const float* part, int nparts, long n, float* out) {
        MbSumJob& jb = jobs.j[jobs.count++];
        jb.part = part;
        jb.out = out;
        jb.n = (int)n;
        jb.nparts = nparts;
        jb.blk0 = jobs.blocks;
        jb.pad = 0;
        jobs.blocks += (int)((n + 63) / 64);
    }
    void flush(hipStream_t s) {
        if (jobs.count == 0) return;
        hipLaunchKernelGGL(sum_jobs_kernel, dim3((unsigned)jobs.blocks), dim3(256), 0, s, jobs);
        jobs.count = 0;
        jobs.blocks = 0;
    }
};

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
    unsigned* counters() const { return reinterpret_cast<unsigned*>(ws + p.counters); }
    Arrive arrive() const { return Arrive{ws + p.part, ws + p.part2, counters(), counters() + MB_CBLOCKS * MB_R2}; }
};

StemGeo stem_geo(const Ctx& c) {
    const Geo& g0 = c.p.g[0];
    const Geo& g1 = c.p.g[1];
    return StemGeo{c.B, g0.ho, g0.win, g0.wo, g0.wy, g1.ho, g1.wo};
}

FinFwd fin_fwd(const Ctx& c, int k, float* buffers) {
    const HowlMbLayer& l = c.n->layers[k];
    return FinFwd{c.params + l.gamma_off, c.params + l.beta_off, buffers + l.rmean_off, buffers + l.rvar_off,
                  c.ws + c.p.ss[k], (double)c.p.g[k].mz};
}
FinBwd fin_bwd(const Ctx& c, int j, float* grads) {
    const HowlMbLayer& l = c.n->layers[j];
    return FinBwd{c.ws + c.p.ss[j], grads + l.gamma_off, grads + l.beta_off, c.ws + c.p.bc[j], (double)c.p.g[j].mz};
}
// the epilogue of the kernel that produces layer j's incoming gradient
EpiBwd epi_bwd(const Ctx& c, int j) {
    const int nl = (int)c.n->layers.size();
    const float* addend = nullptr;
    for (int k2 = j + 1; k2 < nl; ++k2)
        if (c.n->layers[k2].res_src == j) addend = c.ws + c.p.gr[k2];   // y_k2 = bn(z_k2) + y_j, no activation: dy_k2 = g_k2
    return EpiBwd{addend, c.ws + c.p.z[j], c.ws + c.p.ss[j], c.ws + c.p.gr[j], c.n->layers[j].act};
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
    HOWL_REQUIRE(nl <= 64, "howl_mobilenet_fwd: layer table too long");
    HOWL_REQUIRE(c.p.g[0].mz * 3 < (1L << 31), "howl_mobilenet_fwd: batch too large for the stem kernels' 32-bit pixel indices");
    Arrive arr = c.arrive();
    if (!training) arr.part1 = nullptr;     // no statistics: ss comes from the running estimates
    hipMemsetAsync(c.counters(), 0, MB_COUNTERS * sizeof(unsigned), stream);
    if (!training) {
        EvalJobs jobs{};
        for (int k = 0; k < nl; ++k) {
            const HowlMbLayer& l = c.n->layers[k];
            jobs.c[k] = l.cout;
            jobs.gamma_off[k] = l.gamma_off;
            jobs.rmean_off[k] = l.rmean_off;
            jobs.ss_off[k] = (long long)c.p.ss[k];
        }
        hipLaunchKernelGGL(bn_eval_ss_kernel, dim3((MB_MAXC + 255) / 256, nl), dim3(256), 0, stream, jobs, params,
                           (const float*)buffers, c.ws);
    }
    for (int k = 0; k < nl; ++k) {
        const HowlMbLayer& l = c.n->layers[k];
        const Geo& g = c.p.g[k];
        float* z = c.ws + c.p.z[k];
        // the layer's input: y_{k-1} where it exists, else z_{k-1} with the BatchNorm + ReLU6 of layer k-1 applied on load
        const bool in_mat = k == 0 || materialized(c.n->layers[k - 1]);   // (the stem kernels find their own inputs)
        const float* in = k == 0 ? x : (in_mat ? c.ws + c.p.y[k - 1] : c.ws + c.p.z[k - 1]);
        const float* ss_in = in_mat ? nullptr : c.ws + c.p.ss[k - 1];
        const FinFwd fin = fin_fwd(c, k, buffers);
        if (l.kind == MB_PW) {
            // layers whose 64 x 64 tiles would leave CUs idle (the late, deep-reduction bottleneck layers: 96-192 tiles) take
            // 32 x 32 tiles: four times the blocks, a quarter of the MFMA chain per step
            const int tile = pw_tile(g.mz, l.cout);
            const int tc = tile == 32 ? env_int("HOWL_MB_FWD_TC", 32) : 64;    // tile columns (rows = `tile`)
            int tpb, rb;
            pw_rows(g.mz, (l.cout + tc - 1) / tc, &tpb, &rb, tile);
            const dim3 grid((l.cout + tc - 1) / tc, rb);
            HowlProfScope prof("mb_conv", stream, 4.0 * (double)g.mz * (l.cin + l.cout));
            const HowlMbLayer& lp = c.n->layers[k - 1];
            const float* res = nullptr;
            float* y_out = nullptr;
            const float* src = in;
            const float* ssp = ss_in;
            if (in_mat) {     // y_{k-1} = bn(z_{k-1}) (+ y_res) is built here, on load, and stored for its later readers
                src = c.ws + c.p.z[k - 1];
                ssp = c.ws + c.p.ss[k - 1];
                res = lp.res_src >= 0 ? c.ws + c.p.y[lp.res_src] : nullptr;
                y_out = c.ws + c.p.y[k - 1];
            }
#define HOWL_PW_FWD(XF)                                                                                                   \
    do {                                                                                                                  \
        if (tile == 64)                                                                                                      \
            hipLaunchKernelGGL((pw_fwd_kernel<XF, 2, 2>), grid, dim3(256), 0, stream, src, ssp, params + l.w_off, res, \
                               y_out, (int)g.mz, l.cout, l.cin, tpb, z, arr, fin);                                           \
        else if (tc == 64)                                                                                                   \
            hipLaunchKernelGGL((pw_fwd_kernel<XF, 1, 2>), grid, dim3(256), 0, stream, src, ssp, params + l.w_off, res, \
                               y_out, (int)g.mz, l.cout, l.cin, tpb, z, arr, fin);                                           \
        else                                                                                                                 \
            hipLaunchKernelGGL((pw_fwd_kernel<XF, 1, 1>), grid, dim3(256), 0, stream, src, ssp, params + l.w_off, res, \
                               y_out, (int)g.mz, l.cout, l.cin, tpb, z, arr, fin);                                           \
    } while (0)
            if (!in_mat) HOWL_PW_FWD(PW_IN_RELU6);
            else HOWL_PW_FWD(PW_IN_LINEAR);
#undef HOWL_PW_FWD
        } else if (l.kind == MB_DW) {
            int rpc;
            const int nrows = B * g.ho;
            const int chunks = row_chunks(nrows, 4, &rpc);
            const dim3 grid((l.cout + 63) / 64, chunks);
            HowlProfScope prof("mb_conv", stream, 4.0 * ((double)B * g.hin * g.win + (double)g.mz) * l.cout);
            if (l.stride == 1)
                hipLaunchKernelGGL(dw_fwd_kernel<1>, grid, dim3(256), 0, stream, in, ss_in, params + l.w_off, g.hin, g.win, l.cin,
                                   g.ho, g.wo, nrows, rpc, z, arr, fin);
            else
                hipLaunchKernelGGL(dw_fwd_kernel<2>, grid, dim3(256), 0, stream, in, ss_in, params + l.w_off, g.hin, g.win, l.cin,
                                   g.ho, g.wo, nrows, rpc, z, arr, fin);
        } else if (k == 0) {
            int ppb;
            const int chunks = row_chunks(g.mz, 1024, &ppb);
            hipLaunchKernelGGL(stem0_fwd_kernel, dim3(1, chunks), dim3(256), 0, stream, x, sb, sm, st, params + l.w_off,
                               params + l.b_off, stem_geo(c), g.mz, ppb, z, arr, fin);
        } else {
            hipLaunchKernelGGL(stem1_fwd_kernel, dim3((unsigned)((g.mz + 255) / 256)), dim3(256), 0, stream,
                               (const float*)(c.ws + c.p.z[0]), (const float*)(c.ws + c.p.ss[0]), params + l.w_off, stem_geo(c),
                               g.mz, z);
            if (training) {
                int rpc;
                const int cp = col_pack(l.cout);
                const int chunks = row_chunks(g.mz, 64 * (64 / cp), &rpc);
                hipLaunchKernelGGL(col_stats_kernel, dim3(1, chunks), dim3(256), 0, stream, (const float*)z, g.mz, l.cout, cp,
                                   (long)rpc, arr, fin);
            }
        }
    }
    const Geo& gl = c.p.g[nl - 1];
    const float* wc = params + c.n->feature_params;
    const float* bcl = wc + (size_t)num_labels * MB_LAST;
    static_assert(MB_LAST <= 256 * POOL_CPT, "pool_classify_kernel: channels per block");
    hipLaunchKernelGGL(pool_classify_kernel, dim3((unsigned)B), dim3(256), 0, stream, (const float*)(c.ws + c.p.z[nl - 1]),
                       (const float*)(c.ws + c.p.ss[nl - 1]), gl.hy * gl.wy, MB_LAST, training ? drop_mask : (const float*)nullptr,
                       drop_scale, wc, bcl, num_labels, c.ws + c.p.pooled, c.ws + c.p.pooled_d, logits);
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
    const Arrive arr = c.arrive();
    hipMemsetAsync(c.counters(), 0, MB_COUNTERS * sizeof(unsigned), stream);
    // One launch per layer: the data gradient of layer k (which also reduces the BatchNorm backward of the layer in front,
    // so layer k-1's dz exists -- as g, z, bc -- the moment the launch ends) and the weight gradient of layer k share a grid.
    JobList jobs;     // every weight gradient's slab sum, one launch at the end
    SlabSums head;    // classifier
    // classifier: logits = pooled_d W^T + b
    const float* wc = params + c.n->feature_params;
    float* gwc = grads + c.n->feature_params;
    float* gbc = gwc + (size_t)num_labels * MB_LAST;
    {
        // gradient of the pooled features -> g, bc of the last layer
        const int L = nl - 1;
        const Geo& gl = c.p.g[L];
        int upc;
        const int chunks = row_chunks(B, 4, &upc);
        hipLaunchKernelGGL(avgpool_bwd_reduce_kernel, dim3((MB_LAST + 63) / 64, chunks), dim3(256), 0, stream,
                           dlogits, wc, num_labels, drop_mask, drop_scale, gl.hy * gl.wy, MB_LAST, B, upc, epi_bwd(c, L), arr,
                           fin_bwd(c, L, grads));
    }
    wgrad_gemm(stream, dlogits, lin(num_labels), num_labels, c.ws + c.p.pooled_d, lin(MB_LAST), MB_LAST, B,
               c.ws + c.p.head_scratch, gwc, 64, 512, &head);
    colsum(stream, dlogits, lin(num_labels), B, num_labels, c.ws + c.p.bias_scratch, gbc, nullptr, 64, 256, &head);
    for (int k = nl - 1; k >= 2; --k) {
        const HowlMbLayer& l = c.n->layers[k];
        const Geo& g = c.p.g[k];
        const int j = k - 1;
        const HowlMbLayer& lj = c.n->layers[j];
        const bool in_mat = materialized(lj);
        const float* in = in_mat ? c.ws + c.p.y[j] : c.ws + c.p.z[j];
        const float* ss_in = in_mat ? nullptr : c.ws + c.p.ss[j];
        float* slab = c.ws + c.p.slab[k];
        if (l.kind == MB_PW) {
            PwBwd a{};
            a.g = c.ws + c.p.gr[k];
            a.zk = c.ws + c.p.z[k];
            a.bc = c.ws + c.p.bc[k];
            a.w = params + l.w_off;
            a.in = in;
            a.ss_in = ss_in;
            a.M = (int)g.mz;
            a.N = l.cout;
            a.C = l.cin;
            const int tile = pw_tile(g.mz, l.cin);     // the data gradient's output is (M x cin)
            const int tc = tile == 32 ? env_int("HOWL_MB_BWD_TC", 64) : 64;
            a.d_cx = (l.cin + tc - 1) / tc;
            pw_rows(g.mz, a.d_cx, &a.tiles_per_block, &a.d_ry, tile);
            a.rows_per_split = pw_wgrad_rows_per_split(g.mz, l.cout, l.cin);
            a.w_cx = (l.cin + GT - 1) / GT;
            a.w_ny = (l.cout + GT - 1) / GT;
            a.w_nz = (int)((g.mz + a.rows_per_split - 1) / a.rows_per_split);
            a.e = epi_bwd(c, j);
            a.arr = arr;
            a.fin = fin_bwd(c, j, grads);
            a.slabs = slab;
            const unsigned blocks = (unsigned)(a.d_cx * a.d_ry + a.w_cx * a.w_ny * a.w_nz);
            HowlProfScope prof("mb_conv", stream, 4.0 * (double)g.mz * (4.0 * l.cout + 4.0 * l.cin));
#define HOWL_PW_BWD(XF)                                                                                     \
    do {                                                                                                    \
        if (tile == 64) hipLaunchKernelGGL((pw_bwd_kernel<XF, 2, 2>), dim3(blocks), dim3(256), 0, stream, a);       \
        else if (tc == 64) hipLaunchKernelGGL((pw_bwd_kernel<XF, 1, 2>), dim3(blocks), dim3(256), 0, stream, a);    \
        else hipLaunchKernelGGL((pw_bwd_kernel<XF, 1, 1>), dim3(blocks), dim3(256), 0, stream, a);                  \
    } while (0)
            if (ss_in != nullptr) HOWL_PW_BWD(true);
            else HOWL_PW_BWD(false);
#undef HOWL_PW_BWD
            jobs.add(slab, a.w_nz, (long)l.cout * l.cin, grads + l.w_off);
        } else {   // depthwise (layers 0 and 1 are the only dense ones)
            DwBwd a{};
            a.g = c.ws + c.p.gr[k];
            a.zk = c.ws + c.p.z[k];
            a.bc = c.ws + c.p.bc[k];
            a.w = params + l.w_off;
            a.x = in;
            a.ss_in = ss_in;
            a.H = g.hin;
            a.W = g.win;
            a.C = l.cin;
            a.Ho = g.ho;
            a.Wo = g.wo;
            a.ncb = (l.cout + 63) / 64;
            a.d_rows = B * g.hin;
            a.d_chunks = row_chunks(a.d_rows, 4, &a.d_rpc);
            a.w_rows = B * g.ho;
            a.w_chunks = row_chunks(a.w_rows, 8, &a.w_rpc);
            a.e = epi_bwd(c, j);
            a.arr = arr;
            a.fin = fin_bwd(c, j, grads);
            a.slabs = slab;
            const unsigned blocks = (unsigned)(a.ncb * (a.d_chunks + a.w_chunks));
            HowlProfScope prof("mb_conv", stream, 4.0 * (4.0 * (double)g.mz + 3.0 * (double)B * g.hin * g.win) * l.cout);
            if (l.stride == 1)
                hipLaunchKernelGGL(dw_bwd_kernel<1>, dim3(blocks), dim3(256), 0, stream, a);
            else
                hipLaunchKernelGGL(dw_bwd_kernel<2>, dim3(blocks), dim3(256), 0, stream, a);
            jobs.add(slab, a.w_chunks, (long)l.cout * 9, grads + l.w_off);
        }
    }
    // ---- stem ---------------------------------------------------------------------------------------------------------
    {
        const HowlMbLayer& l0 = c.n->layers[0];
        const HowlMbLayer& l1 = c.n->layers[1];
        const Geo& g0 = c.p.g[0];
        const Geo& g1 = c.p.g[1];
        const StemGeo sg = stem_geo(c);
        float* dz1 = c.ws + c.p.dz;
        const long t1 = g1.mz * l1.cout;
        hipLaunchKernelGGL(dz_apply_kernel, dim3(flat_grid(t1)), dim3(256), 0, stream, (const float*)(c.ws + c.p.gr[1]),
                           (const float*)(c.ws + c.p.z[1]), (const float*)(c.ws + c.p.bc[1]), l1.cout, t1, dz1);
        Stem1Bwd a{};
        a.dz1 = dz1;
        a.z0 = c.ws + c.p.z[0];
        a.ss0 = c.ws + c.p.ss[0];
        a.w = params + l1.w_off;
        a.s = sg;
        a.px_per_wblock = stem_px_per_block(g1.mz);
        a.wblocks = stem_blocks(g1.mz);
        const long per_class = (long)B * ((sg.H + 1) / 2) * ((sg.Wp + 1) / 2);
        a.px_per_thread = (int)((per_class + 256L * (MB_R / 4) - 1) / (256L * (MB_R / 4)));
        a.cls_blocks = (int)((per_class + 256L * a.px_per_thread - 1) / (256L * a.px_per_thread));
        a.slabs = c.ws + c.p.slab[1];
        a.g0 = c.ws + c.p.gr[0];
        a.arr = arr;
        a.fin = fin_bwd(c, 0, grads);
        hipLaunchKernelGGL(stem1_bwd_kernel, dim3((unsigned)(a.wblocks + 4 * a.cls_blocks)), dim3(256), 0, stream, a);
        jobs.add(a.slabs, a.wblocks, (long)l1.cout * l1.cin * 9, grads + l1.w_off);
        const int nb0 = stem_blocks(g0.mz);
        hipLaunchKernelGGL(stem0_bwd_kernel, dim3((unsigned)nb0), dim3(256), 0, stream, x, sb, sm, st,
                           (const float*)(c.ws + c.p.gr[0]), (const float*)(c.ws + c.p.z[0]), (const float*)(c.ws + c.p.bc[0]), sg,
                           g0.mz, stem_px_per_block(g0.mz), c.ws + c.p.slab[0], c.ws + c.p.slab0b);
        jobs.add(c.ws + c.p.slab[0], nb0, (long)l0.cout * l0.cin * 9, grads + l0.w_off);
        jobs.add(c.ws + c.p.slab0b, nb0, (long)l0.cout, grads + l0.b_off);
    }
    if (!head.flush(stream)) return HOWL_E_ARG;
    jobs.flush(stream);
    HOWL_CHECK_LAUNCH("howl_mobilenet_bwd");
    return HOWL_OK;
}

}  // extern "C"
