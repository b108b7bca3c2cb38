"""Ablation builds of the 64 x 64 vector GEMM (tools only; see tools/lstm_variants.py): what bounds the six GEMMs of the seq-lstm step?
   python tools/gemm_variants.py build | run"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
HDR = ROOT / "howl_amd" / "csrc" / "howl_gemm.hip.h"
OUT = ROOT / "build" / "diag"
VEC_STORE = "                if (m < M && n < N) cz[(long)m * c_ms + n] = acc[i][j][r];\n            }\n        }\n}\n\n// deterministic sum of `nparts` slabs"
WG_STORE = "                if (m < M && n < N) pz[(long)m * N + n] = acc[i][j][r];"
EDITS = {
    "gbase": [],
    "wnostore": [(WG_STORE, WG_STORE.replace("if (m < M && n < N)", "if (m < M && n < N && acc[i][j][r] == 123.456f)"))],
    "wnomfma": [("        multiply(0);\n        stage(dst1, v1, k0 + WG_K);", "        acc[0][0][0] += As[0][tid] * Bs[0][tid & 255];\n        stage(dst1, v1, k0 + WG_K);"),
                ("        multiply(1);\n        stage(dst0, v0, k0 + 2 * WG_K);", "        acc[0][0][1] += As[1][tid] * Bs[1][tid & 255];\n        stage(dst0, v0, k0 + 2 * WG_K);")],
    "gnostore": [(VEC_STORE, VEC_STORE.replace("if (m < M && n < N)", "if (m < M && n < N && acc[i][j][r] == 123.456f)"))],
    "gnomfma": [("        __builtin_amdgcn_sched_barrier(0);   // keep the requests in front of the MFMAs (the scheduler sinks them otherwise)\n        multiply();\n        __syncthreads();\n        if (k0 + GK >= kend) break;",
                 "        __builtin_amdgcn_sched_barrier(0);\n        acc[0][0][0] += As[tid] * Bs[tid];\n        __syncthreads();\n        if (k0 + GK >= kend) break;"),
                ("        __builtin_amdgcn_sched_barrier(0);\n        multiply();\n        __syncthreads();\n    }\n    float* cz = c + (long)blockIdx.z * c_split_stride;\n    // Bias and ReLU are applied to all 16 results",
                 "        __builtin_amdgcn_sched_barrier(0);\n        acc[0][0][0] += As[tid] * Bs[tid];\n        __syncthreads();\n    }\n    float* cz = c + (long)blockIdx.z * c_split_stride;\n    // Bias and ReLU are applied to all 16 results")],
}


def build():
    OUT.mkdir(parents=True, exist_ok=True)
    text = HDR.read_text()
    objs = [str(p) for p in sorted((ROOT / "build" / "obj").glob("*.o")) if p.name != "lstm.o"]
    try:
        for name, edits in EDITS.items():
            t = text
            for old, new in edits:
                assert t.count(old) == 1, (name, old[:60], t.count(old))
                t = t.replace(old, new)
            HDR.write_text(t)
            obj = OUT / f"lstm_{name}.o"
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w", "-c",
                            str(ROOT / "howl_amd" / "csrc" / "lstm.hip"), "-o", str(obj)], check=True)
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(OUT / f"libhowl_{name}.so"),
                            str(obj)] + objs, check=True)
            print("built", name, flush=True)
    finally:
        HDR.write_text(text)


def run():
    os.chdir("/tmp")
    for name in EDITS:
        env = dict(os.environ, HOWL_HIP_LIBRARY=str(OUT / f"libhowl_{name}.so"), TMPDIR="/tmp", NUM_MELS="40")
        d = ROOT / "gpurun_out" / "gemmvar" / name
        subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", str(d), "-o", "t", "--", sys.executable,
                        str(ROOT / "bench.py"), "--config", "c4", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-roofline"],
                       env=env, capture_output=True, text=True)
        f = next(d.rglob("*kernel_trace.csv"))
        r = subprocess.run([sys.executable, str(ROOT / "tools" / "step_timeline.py"), str(f)], capture_output=True, text=True)
        print("==", name)
        print("\n".join(l for l in r.stdout.splitlines() if "gemm" in l or "wgrad" in l or "step:" in l), flush=True)


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
