"""Measurement tool (not product): builds an instrumented COPY of the conv3x3 MFMA kernel (s_memtime stamps per
phase for the 12 waves of workgroup 0) and prints the timeline.  Usage on the GPU box: python tools/probe_conv.py"""
import ctypes
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
src = (ROOT / "howl_amd" / "csrc" / "res8.hip").read_text()
src = src.replace('#include "howl_common.hip.h"', f'#include "{ROOT}/howl_amd/csrc/howl_common.hip.h"')
src = src.replace('#include "../../include/howl_hip.h"', f'#include "{ROOT}/include/howl_hip.h"')


def sub(old, new, count=1):
    global src
    assert src.count(old) >= 1, old
    src = src.replace(old, new, count)


# probe storage + helper
sub("namespace {\n\nstruct HowlPtrs6", """__device__ long long g_probe[12 * 128];
#define STAMP(i) do { if (blockIdx.x == g_probe_block && lane == 0 && (i) < 128) g_probe[wave * 128 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ int g_probe_block = 0;
namespace {

struct HowlPtrs6""")
# stamps inside the MODE-templated kernel
sub("    for (int i = tid; i < 3 * KSTEPS * 16; i += CONV_THREADS)\n        reinterpret_cast<float4*>(wl)[i]", "    int pi = 0;\n    STAMP(pi);\n    ++pi;\n    for (int i = tid; i < 3 * KSTEPS * 16; i += CONV_THREADS)\n        reinterpret_cast<float4*>(wl)[i]")
sub("    __syncthreads();  // weights, zero fill and stats visible before the first stage\n", "    STAMP(pi);\n    ++pi;\n    __syncthreads();  // weights, zero fill and stats visible before the first stage\n    STAMP(pi);\n    ++pi;\n")
sub("        stage_tile(pre, pk, tile, lmean, lrstd, affine);\n        __syncthreads();\n        const int bn = b + gridDim.x;\n        if (bn < B) prefetch_tile(pre, in + (size_t)bn * NMAP * P, n2, tid);\n",
    "        STAMP(pi);\n        ++pi;\n        stage_tile(pre, pk, tile, lmean, lrstd, affine);\n        STAMP(pi);\n        ++pi;\n        __syncthreads();\n        STAMP(pi);\n        ++pi;\n        const int bn = b + gridDim.x;\n        if (bn < B) prefetch_tile(pre, in + (size_t)bn * NMAP * P, n2, tid);\n        STAMP(pi);\n        ++pi;\n")
sub("        __syncthreads();  // single tile buffer: every wave is done reading before the next utterance is staged",
    "        STAMP(pi);\n        ++pi;\n        __syncthreads();  // single tile buffer: every wave is done reading before the next utterance is staged\n        STAMP(pi);\n        ++pi;")
# ---- wgrad kernel stamps (second STAMP family writes to g_probe2)
sub("__device__ int g_probe_block = 0;", """__device__ int g_probe_block = 0;
__device__ long long g_probe2[12 * 128];
#define STAMP2(i) do { if (blockIdx.x == g_probe_block && lane == 0 && (i) < 128) g_probe2[wave * 128 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)""")
sub("    zero_lds(lds, 2 * TF, tid, CONV_THREADS);\n    if (tid < CP) {\n        lmean[tid] = affine ? in_stats[tid] : 0.0f;\n        lrstd[tid] = affine ? in_stats[CP + tid] : 1.0f;\n    }\n    // this wave's N tiles",
    "    int pi = 0;\n    STAMP2(pi);\n        ++pi;\n    zero_lds(lds, 2 * TF, tid, CONV_THREADS);\n    if (tid < CP) {\n        lmean[tid] = affine ? in_stats[tid] : 0.0f;\n        lrstd[tid] = affine ? in_stats[CP + tid] : 1.0f;\n    }\n    // this wave's N tiles")
sub("        stage_tile(pz, pk, tz, lmean, lrstd, false);\n        stage_tile(px, pk, tx, lmean, lrstd, affine);\n        __syncthreads();\n",
    "        STAMP2(pi);\n        ++pi;\n        stage_tile(pz, pk, tz, lmean, lrstd, false);\n        stage_tile(px, pk, tx, lmean, lrstd, affine);\n        STAMP2(pi);\n        ++pi;\n        __syncthreads();\n        STAMP2(pi);\n        ++pi;\n")
sub("        // K loop over positions, 4 per MFMA", "        STAMP2(pi);\n        ++pi;\n        // K loop over positions, 4 per MFMA")
sub("        __syncthreads();  // single-buffered tiles: everyone done before the next stage overwrites them",
    "        STAMP2(pi);\n        ++pi;\n        __syncthreads();  // single-buffered tiles: everyone done before the next stage overwrites them\n        STAMP2(pi);\n        ++pi;")
src += r'''
extern "C" int probe_wgrad(int B, int H, long long* host_out, float* ms_out) {
    const int P = H * PW;
    const size_t act = (size_t)B * NMAP * P;
    float *dz, *sp, *part, *stats;
    hipMalloc(&dz, act * 4); hipMalloc(&sp, act * 4); hipMalloc(&part, (size_t)256 * CP * 432 * 4); hipMalloc(&stats, 2 * CP * 4);
    hipMemset(dz, 0, act * 4); hipMemset(sp, 0, act * 4); hipMemset(stats, 0, 2 * CP * 4);
    const size_t lc = conv_lds_bytes(H);
    hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lc);
    int G = B < 256 ? B : 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(wgrad_mfma_kernel, dim3(G), dim3(CONV_THREADS), lc, 0, dz, sp, stats, part, B, H);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(ms_out, e0, e1);
    }
    hipDeviceSynchronize();
    int rc = (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_probe2), sizeof(long long) * 12 * 128);
    return rc * 1000 + (int)hipGetLastError();
}
extern "C" int probe_run(int B, int H, int with_res, long long* host_out, float* ms_out) {
    const int P = H * PW;
    const size_t act = (size_t)B * NMAP * P;
    float *in, *out, *res, *y, *wp, *part, *stats;
    hipMalloc(&in, act * 4); hipMalloc(&out, act * 4); hipMalloc(&res, act * 4); hipMalloc(&y, act * 4);
    hipMalloc(&wp, 3 * KSTEPS * 64 * 4); hipMalloc(&part, 256 * 2 * CP * 4); hipMalloc(&stats, 2 * CP * 4);
    hipMemset(in, 0, act * 4); hipMemset(res, 0, act * 4); hipMemset(wp, 0, 3 * KSTEPS * 64 * 4); hipMemset(stats, 0, 2 * CP * 4);
    const size_t lc = conv_lds_bytes(H);
    hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_mfma_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lc);
    int G = B < 256 ? B : 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(conv3x3_mfma_kernel<0>, dim3(G), dim3(CONV_THREADS), lc, 0, in, stats, wp, with_res ? res : nullptr,
                           with_res ? y : nullptr, out, (const float*)nullptr, (const float*)nullptr, part, B, H);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(ms_out, e0, e1);
    }
    hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_probe), sizeof(long long) * 12 * 128);
    return (int)hipGetLastError();
}
'''
out = Path("/tmp/probe_conv.hip")
out.write_text(src)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
                str(out), str(ROOT / "howl_amd/csrc/capi.hip"), "-o", "/tmp/libprobe.so"], check=True)
if "--build-only" in sys.argv:
    sys.exit(0)
lib = ctypes.CDLL("/tmp/libprobe.so")
buf = (ctypes.c_longlong * (12 * 128))()
ms = ctypes.c_float()
def show(tag, nshow=40):
    t = [[buf[w * 128 + i] for i in range(128)] for w in range(12)]
    t0 = min([t[w][0] for w in range(12) if t[w][0]] or [0])
    for w in (0, 4, 8, 3, 7, 11):
        ev = [x - t0 for x in t[w] if x]
        d = [ev[0]] + [ev[i] - ev[i - 1] for i in range(1, len(ev))]
        print(f"{tag} wave {w:2d} deltas(x100cyc): " + " ".join(f"{e / 100:.0f}" for e in d[:nshow]))


for B in (512, 2048):
    rc = lib.probe_wgrad(B, 27, buf, ctypes.byref(ms))
    print(f"\n=== wgrad B={B} rc={rc} kernel {ms.value * 1e3:.1f} us; units: 100 shader cycles; "
          "events: start, [stage-begin, stage-end, barrier-end, kloop-begin, kloop-end, barrier-end]*")
    show("wgrad")
for B, with_res in ((512, 0), (512, 1), (2048, 0)):
    rc = lib.probe_run(B, 27, with_res, buf, ctypes.byref(ms))
    print(f"\n=== conv B={B} res={with_res} rc={rc} kernel {ms.value * 1e3:.1f} us; units: 100 shader cycles; events: start, "
          "setup-end, barrier-end, [stage-begin, stage-end, barrier-end, prefetch-issued, compute+epilogue-end, barrier-end]*")
    show("conv")
