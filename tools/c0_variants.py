"""Ablation builds of conv0's weight-gradient kernel (tools only; see tools/lstm_variants.py): load phase vs compute phase.
   python tools/c0_variants.py build | run"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "howl_amd" / "csrc" / "res8.hip"
OUT = ROOT / "build" / "diag"
EDITS = {
    "c0base": [],
    "c0nocompute": [("        for (int cell = c0 + wave; cell < c1; cell += C0W_THREADS / 64) {", "        for (int cell = c0 + wave; cell < c1 && B < 0; cell += C0W_THREADS / 64) {")],
    "c0noload": [("        for (int i0 = tid; i0 < NMAP * nc; i0 += 8 * C0W_THREADS) {   // bulk, 8 loads in flight per thread: the slice's cells of",
                  "        for (int i0 = tid; i0 < NMAP * nc && B < 0; i0 += 8 * C0W_THREADS) {   // bulk, 8 loads in flight per thread: the slice's cells of")],
}


def build():
    OUT.mkdir(parents=True, exist_ok=True)
    text = SRC.read_text()
    objs = [str(p) for p in sorted((ROOT / "build" / "obj").glob("*.o")) if p.name != "res8.o"]
    for name, edits in EDITS.items():
        t = text
        for old, new in edits:
            assert t.count(old) == 1, (name, old[:60], t.count(old))
            t = t.replace(old, new)
        tmp = SRC.parent / f"_diag_{name}.hip"
        tmp.write_text(t)
        try:
            obj = OUT / f"res8_{name}.o"
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w", "-c", str(tmp), "-o", str(obj)], check=True)
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(OUT / f"libhowl_{name}.so"), str(obj)] + objs, check=True)
        finally:
            tmp.unlink()
        print("built", name, flush=True)


def run():
    os.chdir("/tmp")
    for name in EDITS:
        env = dict(os.environ, HOWL_HIP_LIBRARY=str(OUT / f"libhowl_{name}.so"), TMPDIR="/tmp", NUM_MELS="40")
        d = ROOT / "gpurun_out" / "c0var" / name
        subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", str(d), "-o", "t", "--", sys.executable,
                        str(ROOT / "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-roofline"],
                       env=env, capture_output=True, text=True)
        f = next(d.rglob("*kernel_stats.csv"))
        for line in f.read_text().splitlines():
            if "conv0_wgrad" in line:
                parts = line.split(",")
                print(name, "conv0_wgrad avg_ns", parts[-5], flush=True)


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
