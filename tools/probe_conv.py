"""Per-phase timeline of conv3x3_mfma_kernel<0> (workgroup 0, one s_memtime stamp per wave and phase; diagnostic build with
-DHOWL_DIAG_PROBE).  Runs on the GPU box:  python tools/probe_conv.py [batch]"""
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
from howl_amd import lib as hlib
from howl_amd.data.transform.operator import ZmuvTransform
from howl_amd.data.transform.transform import StandardAudioTransform
from howl_amd.model import RegisteredModel
from howl_amd.utils.synth import res8_closed_form_state, synthetic_pcm
B = int(sys.argv[1]); dev = torch.device("cuda:0"); lb = hlib.get()
pcm = synthetic_pcm(B, 16000).to(dev)
std = StandardAudioTransform().to(dev).eval(); zmuv = ZmuvTransform().to(dev); zmuv.update(std(pcm[:8]))
model = RegisteredModel.find_registered_class("res8")(12).to(dev)
model.load_state_dict(res8_closed_form_state(12), strict=False); model.train()
feat = std.log_mel_for_model(pcm, zmuv)
for _ in range(3): model._launch_forward(feat)
buf = torch.zeros(12 * 64, dtype=torch.int64, device=dev)
lb.cdll.howl_diag_set_probe.argtypes = [ctypes.c_void_p]
assert lb.cdll.howl_diag_set_probe(buf.data_ptr()) == 0
model._launch_forward(feat)           # the last 3x3 launch (layer 6: statistics fold + residual) leaves its stamps
torch.cuda.synchronize()
t = buf.cpu().view(12, 64)
t0 = int(t[:, 0].min())
print("ticks (~shader cycles) relative to the first wave's entry; columns: entry requested folded zeroed weights slots barrier done")
for w in (0, 4, 8, 3, 7, 11):
    row = [int(v) - t0 for v in t[w] if int(v) != 0]
    d = [row[0]] + [row[i] - row[i - 1] for i in range(1, len(row))]
    print("wave %%2d deltas:" %% w, " ".join("%%6d" %% v for v in d), "| total", row[-1])
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
    env = dict(os.environ, HOWL_HIP_LIBRARY=str(so))
    for B in ([int(sys.argv[1])] if len(sys.argv) > 1 else [512]):
        print(f"== B = {B}")
        r = subprocess.run([sys.executable, "-c", CHILD, str(B)], env=env, capture_output=True, text=True, timeout=300)
        print(r.stdout.strip() or r.stderr[-1500:], flush=True)


if __name__ == "__main__":
    main()
