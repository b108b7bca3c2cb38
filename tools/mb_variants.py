"""Ablation builds of the MobileNet depthwise backward launch (tools only; see tools/lstm_variants.py): which role is the long pole?
   python tools/mb_variants.py build | run"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "howl_amd" / "csrc" / "mobilenet.hip"
OUT = ROOT / "build" / "diag"
EDITS = {
    "mbase": [],
    "mnowgrad": [("        dw_wgrad_body<STRIDE>(p, wb % p.ncb, wb / p.ncb);", "        if (p.C < 0) dw_wgrad_body<STRIDE>(p, wb % p.ncb, wb / p.ncb);")],
    "mdg_nostore": [("                    if (FULL || cok) gjr[(long)iw * C] = gg;", "                    if ((FULL || cok) && gg == 123.456f) gjr[(long)iw * C] = gg;")],
    "mdg_nowin": [("                    v[kh][j] = gb[((long)ohs[kh] * Wo + owc) * C];\n                    vz[kh][j] = zb[((long)ohs[kh] * Wo + owc) * C];",
                   "                    v[kh][j] = ksc + j;\n                    vz[kh][j] = kc1 + kh;")],
    "mdg_nozj": [("            for (int o = 0; o < DW_SEG; ++o) zj[o] = zjr[(long)min(iw0 + o, W - 1) * C];", "            for (int o = 0; o < DW_SEG; ++o) zj[o] = jsc + o;")],
    "mnodgrad": [("        dw_dgrad_body<STRIDE>(p, b % p.ncb, b / p.ncb);", "        if (p.C < 0) dw_dgrad_body<STRIDE>(p, b % p.ncb, b / p.ncb);")],
}


def build():
    OUT.mkdir(parents=True, exist_ok=True)
    text = SRC.read_text()
    objs = [str(p) for p in sorted((ROOT / "build" / "obj").glob("*.o")) if p.name != "mobilenet.o"]
    for name, edits in EDITS.items():
        t = text
        for old, new in edits:
            assert t.count(old) == 1, (name, old[:60], t.count(old))
            t = t.replace(old, new)
        tmp = SRC.parent / f"_diag_{name}.hip"
        tmp.write_text(t)
        try:
            obj = OUT / f"mb_{name}.o"
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w", "-c", str(tmp), "-o", str(obj)], check=True)
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(OUT / f"libhowl_{name}.so"), str(obj)] + objs, check=True)
        finally:
            tmp.unlink()
        print("built", name, flush=True)


def run():
    os.chdir("/tmp")
    for name in EDITS:
        env = dict(os.environ, HOWL_HIP_LIBRARY=str(OUT / f"libhowl_{name}.so"), TMPDIR="/tmp", NUM_MELS="40")
        d = ROOT / "gpurun_out" / "mbvar" / name
        subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", str(d), "-o", "t", "--", sys.executable,
                        str(ROOT / "bench.py"), "--config", "c5", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-roofline"],
                       env=env, capture_output=True, text=True)
        f = next(d.rglob("*kernel_stats.csv"))
        print("==", name)
        for line in f.read_text().splitlines():
            if "dw_bwd" in line:
                parts = line.split(",")
                print("  ", line.split("(")[0][-30:], "calls", parts[-7], "avg_ns", parts[-5], "min", parts[-3], "max", parts[-2], flush=True)


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
