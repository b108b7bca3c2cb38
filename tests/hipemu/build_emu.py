"""Build tests/hipemu/libhowl_emu.so: the kernels under howl_amd/csrc compiled UNMODIFIED for the host with the
hipemu fiber emulator (test infrastructure only; see hip/hip_runtime.h)."""
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
OUT = HERE / "libhowl_emu.so"


def build(verbose=False):
    srcs = sorted((ROOT / "howl_amd" / "csrc").glob("*.hip")) + [HERE / "hipemu.cpp"]
    newest = max(p.stat().st_mtime for p in list((ROOT / "howl_amd" / "csrc").glob("*")) + list(HERE.glob("*.cpp")) +
                 list((HERE / "hip").glob("*")) + list((ROOT / "include").glob("*")))
    if OUT.exists() and OUT.stat().st_mtime > newest:
        return OUT
    cmd = [CLANG, "-std=c++17", "-O2", "-g", "-fPIC", "-shared", "-Wno-unused-value", "-Wno-unknown-attributes",
           f"-I{HERE}", "-o", str(OUT)]
    for s in srcs:
        cmd += ["-x", "c++", str(s)]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(verbose=True))
