# NOTE (round 5): the HOWL_DIAG_* branches this tool compiles were removed from the product kernels (tools/strip_diag.py);
# it builds against the sources of commit 37e3835 (`git worktree add /tmp/howl_r4 37e3835` and run it there).
"""Diagnostic variants of logmel_kernel (results WRONG by design: each removes one part) timed at 512 / 2048 x 1 s.
Runs on the GPU box:  python tools/logmel_variants.py"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "howl_amd" / "csrc"
VARIANTS = {"base": [], "nofft": ["-DHOWL_DIAG_LOGMEL_NOFFT"], "nomel": ["-DHOWL_DIAG_LOGMEL_NOMEL"],
            "noload": ["-DHOWL_DIAG_LOGMEL_NOLOAD"], "nofft_nomel": ["-DHOWL_DIAG_LOGMEL_NOFFT", "-DHOWL_DIAG_LOGMEL_NOMEL"],
            "nothing": ["-DHOWL_DIAG_LOGMEL_NOFFT", "-DHOWL_DIAG_LOGMEL_NOMEL", "-DHOWL_DIAG_LOGMEL_NOLOAD"]}
CHILD = r"""
import os, sys
sys.path.insert(0, %r)
os.environ.setdefault("NUM_MELS", "40")
import torch
from howl_amd import ops
from howl_amd.data.transform.transform import StandardAudioTransform
from howl_amd.utils.synth import synthetic_pcm
dev = torch.device("cuda:0")
std = StandardAudioTransform().to(dev).eval()
fbp = std._standard_fb()
pair = torch.tensor([0.0, 1.0], device=dev)
def bench(B, n=30):
    pcm = synthetic_pcm(B, 16000).to(dev)
    for _ in range(5): ops.logmel(pcm, fbp, 40, pair, layout=1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.logmel(pcm, fbp, 40, pair, layout=1)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(" ".join("B=%%d %%.1f us" %% (B, bench(B)) for B in (256, 512, 1024, 2048)), flush=True)
""" % str(ROOT)


def main():
    out = Path("/tmp/howl_variants")
    out.mkdir(exist_ok=True)
    objs = [str(ROOT / "build" / "obj" / f"{f}.o") for f in ("capi", "ctc", "lstm", "mobilenet", "res8")]
    for name, flags in VARIANTS.items():
        obj, so = out / f"frontend_{name}.o", out / f"libhowl_fe_{name}.so"
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", *flags, "-c",
                            str(CSRC / "frontend.hip"), "-o", str(obj)], capture_output=True, text=True)
        if r.returncode == 0:
            r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", str(obj), *objs, "-o", str(so)], capture_output=True, text=True)
        if r.returncode != 0:
            print(f"{name}: build failed\n{r.stderr[-1500:]}")
            continue
        for waves in ("12", "16"):
            env = dict(os.environ, HOWL_HIP_LIBRARY=str(so), HOWL_LOGMEL_WAVES=waves)
            r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
            print(f"{name:12s} waves={waves}: {r.stdout.strip() or r.stderr[-600:]}", flush=True)


if __name__ == "__main__":
    main()
