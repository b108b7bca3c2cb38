"""Measurement tool (not product): phase timestamps inside conv0_fwd_kernel / conv0_wgrad_kernel (workgroup 0)."""
import ctypes
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
src = (ROOT / "howl_amd" / "csrc" / "res8.hip").read_text()
src = src.replace('#include "howl_common.hip.h"', f'#include "{ROOT}/howl_amd/csrc/howl_common.hip.h"')
src = src.replace('#include "../../include/howl_hip.h"', f'#include "{ROOT}/include/howl_hip.h"')


def sub(old, new):
    global src
    assert src.count(old) >= 1, old
    src = src.replace(old, new, 1)


sub("namespace {\n\nstruct HowlPtrs6", """__device__ long long g_probe[16 * 64];
#define STAMP(i) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (i) < 64) g_probe[(threadIdx.x >> 6) * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
namespace {

struct HowlPtrs6""")
# conv0_fwd: stamps around tile load and compute
a = src.index("void conv0_fwd_kernel(")
seg = src[a:]
seg = seg.replace("    for (int b = blockIdx.x; b < B; b += gridDim.x) {\n        __syncthreads();\n        load_feat_tile(tin, feat, sb, st, sm, b, T, M, tid, C0_THREADS);\n        __syncthreads();",
                  "    int pi = 0;\n    STAMP(pi); ++pi;\n    for (int b = blockIdx.x; b < B; b += gridDim.x) {\n        __syncthreads();\n        STAMP(pi); ++pi;\n        load_feat_tile(tin, feat, sb, st, sm, b, T, M, tid, C0_THREADS);\n        __syncthreads();\n        STAMP(pi); ++pi;", 1)
seg = seg.replace("                if (mask0 != nullptr) mask0[((size_t)b * NMAP + c) * P + p] = (unsigned short)bits;\n            }\n        }\n    }\n}",
                  "                if (mask0 != nullptr) mask0[((size_t)b * NMAP + c) * P + p] = (unsigned short)bits;\n            }\n        }\n        STAMP(pi); ++pi;\n    }\n}", 1)
src = src[:a] + seg
src += r'''
extern "C" int probe_conv0(int B, int T, long long* host_out, float* ms_out) {
    const int H = T / 3, P = H * PW, M = 40;
    float *feat, *w0, *s0; unsigned short* mk;
    hipMalloc(&feat, (size_t)B * T * M * 4); hipMalloc(&w0, 405 * 4); hipMalloc(&s0, (size_t)B * NMAP * P * 4); hipMalloc(&mk, (size_t)B * NMAP * P * 2);
    hipMemset(feat, 0, (size_t)B * T * M * 4); hipMemset(w0, 0, 405 * 4);
    const size_t l0 = conv0_lds_bytes(T, M);
    int G = B < 256 ? B : 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(conv0_fwd_kernel, dim3(G), dim3(C0_THREADS), l0, 0, feat, (long)T * M, (long)M, 1L, w0, s0, mk, B, T, M, H);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(ms_out, e0, e1);
    }
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_probe), sizeof(long long) * 16 * 64);
    return (int)hipGetLastError();
}
'''
out = Path("/tmp/probe_conv0.hip")
out.write_text(src)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
                str(out), str(ROOT / "howl_amd/csrc/capi.hip"), "-o", "/tmp/libprobe0.so"], check=True)
import sys
if "--build-only" in sys.argv:
    sys.exit(0)
lib = ctypes.CDLL("/tmp/libprobe0.so")
buf = (ctypes.c_longlong * (16 * 64))()
ms = ctypes.c_float()
for B in (256, 512, 2048):
    rc = lib.probe_conv0(B, 81, buf, ctypes.byref(ms))
    print(f"=== conv0_fwd B={B} rc={rc} kernel {ms.value * 1e3:.1f} us; events: start, [loop-top, tile-loaded, compute-done]*; deltas x100 cycles")
    for w in (0, 4, 8):
        ev = [buf[w * 64 + i] for i in range(64) if buf[w * 64 + i]]
        d = [0] + [ev[i] - ev[i - 1] for i in range(1, len(ev))]
        print(f"wave {w}: " + " ".join(f"{x / 100:.0f}" for x in d[:26]))
