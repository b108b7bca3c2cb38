"""Build tests/hipemu/libhowl_emu.so: the kernels under howl_amd/csrc compiled UNMODIFIED for the host with the
hipemu fiber emulator (test infrastructure only; see hip/hip_runtime.h).  One object per source (only the stale ones are
recompiled, side by side), then one link."""
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
OUT = HERE / "libhowl_emu.so"
OBJ = ROOT / "build" / "emu_obj"
# The kernels ask for full unrolling wherever their device registers are indexed by a loop variable; the host has no such need,
# and honouring it made one object (res8: 510 emulated MFMA calls per utterance body, a dozen instantiations) a five-minute
# compile.  The thresholds keep `#pragma unroll` to small bodies: 30 s, same results (the emulator runs the same statements).
FLAGS = ["-std=c++17", "-O2", "-fPIC", "-Wno-unused-value", "-Wno-unknown-attributes", "-Wno-pass-failed",
         "-mllvm", "-pragma-unroll-threshold=256", "-mllvm", "-unroll-threshold=64", f"-I{HERE}"]


def build(verbose=False):
    srcs = sorted(p for p in (ROOT / "howl_amd" / "csrc").glob("*.hip") if not p.name.startswith("_")) + [HERE / "hipemu.cpp"]
    hdrs = [p for p in (ROOT / "howl_amd" / "csrc").glob("*") if p.suffix != ".hip"] + list((HERE / "hip").glob("*")) + \
        list((ROOT / "include").glob("*")) + [Path(__file__)]
    hdr_time = max(p.stat().st_mtime for p in hdrs)
    OBJ.mkdir(parents=True, exist_ok=True)
    jobs, objs = [], []
    for s in srcs:
        o = OBJ / (s.stem + ".o")
        objs.append(o)
        if not o.exists() or o.stat().st_mtime < max(s.stat().st_mtime, hdr_time):
            cmd = [CLANG] + FLAGS + ["-c", "-x", "c++", str(s), "-o", str(o)]
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append((s, subprocess.Popen(cmd)))
    failed = [str(s) for s, proc in jobs if proc.wait() != 0]
    if failed:
        raise RuntimeError("hipemu build failed for " + ", ".join(failed))
    if jobs or not OUT.exists() or OUT.stat().st_mtime < max(o.stat().st_mtime for o in objs):
        subprocess.run([CLANG, "-shared", "-fPIC", "-o", str(OUT)] + [str(o) for o in objs], check=True)
    return OUT


if __name__ == "__main__":
    print(build(verbose=True))
