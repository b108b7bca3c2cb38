# NOTE (round 5): the HOWL_DIAG_* branches this tool compiles were removed from the product kernels (tools/strip_diag.py);
# it builds against the sources of commit 37e3835 (`git worktree add /tmp/howl_r4 37e3835` and run it there).
"""Timeline and occupancy probe of logmel_kernel (diagnostic build with -DHOWL_DIAG_PROBE: one s_memtime stamp per wave and
phase of workgroup 0).  Runs on the GPU box:  python tools/probe_logmel.py"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "howl_amd" / "csrc"

CHILD = r"""
import ctypes, os, sys
sys.path.insert(0, %r)
os.environ.setdefault("NUM_MELS", "40")
import torch
from howl_amd import lib as hlib, ops
from howl_amd.data.transform.transform import StandardAudioTransform
from howl_amd.utils.synth import synthetic_pcm
dev = torch.device("cuda:0"); lb = hlib.get()
std = StandardAudioTransform().to(dev).eval()
fbp = std._standard_fb()
pair = torch.tensor([0.0, 1.0], device=dev)
def bench(B, L=16000, n=30):
    pcm = synthetic_pcm(B, L).to(dev)
    for _ in range(5): ops.logmel(pcm, fbp, 40, pair, layout=1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.logmel(pcm, fbp, 40, pair, layout=1)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
mode = sys.argv[1]
if mode == "time":
    print("logmel waves=%%s:" %% os.environ.get("HOWL_LOGMEL_WAVES"),
          " ".join("B=%%d %%.1f us" %% (B, bench(B)) for B in (8, 64, 128, 256, 512, 1024, 2048)), flush=True)
else:
    B = int(sys.argv[2])
    pcm = synthetic_pcm(B, 16000).to(dev)
    for _ in range(3): ops.logmel(pcm, fbp, 40, pair, layout=1)
    buf = torch.zeros(16 * 64, dtype=torch.int64, device=dev)
    lb.cdll.howl_diag_set_probe_fe.argtypes = [ctypes.c_void_p]
    assert lb.cdll.howl_diag_set_probe_fe(buf.data_ptr()) == 0
    ops.logmel(pcm, fbp, 40, pair, layout=1)
    torch.cuda.synchronize()
    t = buf.cpu().view(16, 64)
    first = t[:, 0]; t0 = int(first[first > 0].min()) if (first > 0).any() else 0
    print("B=%%d ticks relative to the first stamp; columns per quad: start transformed contracted stored" %% B)
    for w in range(16):
        row = [int(v) - t0 for v in t[w] if int(v) != 0]
        if not row: continue
        d = [row[i] - row[i - 1] for i in range(1, len(row))]
        print("wave %%2d prologue %%6d" %% (w, row[0]))
        for k in range(0, len(d), 4):
            print("   quad:", " ".join("%%5d" %% v for v in d[k:k + 4]), "| sum", sum(d[k:k + 4]))
        print("   end at", row[-1])
""" % str(ROOT)


def main():
    out = Path("/tmp/howl_variants")
    out.mkdir(exist_ok=True)
    objs = []
    for f in ("capi", "ctc", "frontend", "lstm", "mobilenet", "res8"):
        o = out / f"probe_{f}.o"
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-DHOWL_DIAG_PROBE",
                        "-c", str(CSRC / f"{f}.hip"), "-o", str(o)], check=True)
        objs.append(str(o))
    so = out / "libhowl_probe.so"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", str(so)], check=True)
    for waves in ("12", "16"):
        env = dict(os.environ, HOWL_LOGMEL_WAVES=waves)
        r = subprocess.run([sys.executable, "-c", CHILD, "time"], env=env, capture_output=True, text=True, timeout=300)
        print(r.stdout.strip() or r.stderr[-1500:], flush=True)
    env = dict(os.environ, HOWL_HIP_LIBRARY=str(so))
    for B in (8, 512):
        r = subprocess.run([sys.executable, "-c", CHILD, "probe", str(B)], env=env, capture_output=True, text=True, timeout=300)
        print(r.stdout.strip() or r.stderr[-1500:], flush=True)


if __name__ == "__main__":
    main()
