"""Diagnostic kernel variants: build the library with HOWL_DIAG_* switches (results are WRONG by design -- they remove one
resource from a kernel to show what bounds it) and time the res8 kernels at B=512.  Runs on the GPU box:
    python tools/variants.py [variant ...]
"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "howl_amd" / "csrc"
VARIANTS = {
    "base": [],
    "logmel_nofft": ["-DHOWL_DIAG_LOGMEL_NOFFT"],
    "logmel_nomel": ["-DHOWL_DIAG_LOGMEL_NOMEL"],
    "logmel_noload": ["-DHOWL_DIAG_LOGMEL_NOLOAD"],
    "conv_nolds": ["-DHOWL_DIAG_CONV_NOLDS"],
    "conv_nomfma": ["-DHOWL_DIAG_CONV_NOMFMA"],
    "conv_ntstore": ["-DHOWL_DIAG_CONV_NTSTORE"],
    "conv_nostore": ["-DHOWL_DIAG_CONV_NOSTORE"],
    "conv_nok": ["-DHOWL_DIAG_CONV_NOK"],
    "conv_nowload": ["-DHOWL_DIAG_CONV_NOWLOAD"],
    "conv_noprio": ["-DHOWL_DIAG_CONV_NOPRIO"],
    "conv0_nostore": ["-DHOWL_DIAG_C0_NOSTORE"],
    "conv0_nomfma": ["-DHOWL_DIAG_C0_NOMFMA"],
    "wgrad_nolds": ["-DHOWL_DIAG_WGRAD_NOLDS"],
    "wgrad_nomfma": ["-DHOWL_DIAG_WGRAD_NOMFMA"],
}

CHILD = r"""
import ctypes, os, sys, time
sys.path.insert(0, %r)
os.environ.setdefault("NUM_MELS", "40")
import torch
from howl_amd import lib as hlib
from howl_amd.data.transform.operator import ZmuvTransform
from howl_amd.data.transform.transform import StandardAudioTransform
from howl_amd.model import RegisteredModel
from howl_amd.training.fused import FusedRes8Trainer
from howl_amd.utils.synth import res8_closed_form_state, synthetic_pcm
dev = torch.device("cuda:0"); lb = hlib.get(); B = 512
pcm = synthetic_pcm(B, 16000).to(dev); labels = (torch.arange(B) %% 12).to(dev)
std = StandardAudioTransform().to(dev).eval(); zmuv = ZmuvTransform().to(dev); zmuv.update(std(pcm[:8]))
model = RegisteredModel.find_registered_class("res8")(12).to(dev)
model.load_state_dict(res8_closed_form_state(12), strict=False); model.train()
tr = FusedRes8Trainer(model, std, zmuv, lr=0.0)
for _ in range(5): tr.step(pcm, labels)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): tr.step(pcm, labels)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
lb.call("howl_profile_enable", 1)
for _ in range(10): tr.step(pcm, labels)
torch.cuda.synchronize(); lb.call("howl_profile_enable", 0)
row = ["step %%.3f ms |" %% (dt * 1e3)]
for tag in ("conv3x3_fwd", "conv3x3_dgrad", "wgrad", "logmel", "conv0_fwd"):
    tot, cnt = ctypes.c_double(0), ctypes.c_int(0)
    lb.call("howl_profile_read", tag.encode(), ctypes.byref(tot), ctypes.byref(cnt), 0)
    row.append("%%s %%.1f us" %% (tag, tot.value / max(cnt.value, 1) * 1e3))
print(" ".join(row), flush=True)
""" % str(ROOT)


def main():
    names = sys.argv[1:] or list(VARIANTS)
    out = Path("/tmp/howl_variants")
    out.mkdir(exist_ok=True)
    objs = []
    for f in ("capi", "lstm", "mobilenet", "ctc"):
        o = out / f"{f}.o"
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-c", str(CSRC / f"{f}.hip"), "-o", str(o)],
                       check=True)
        objs.append(str(o))
    for name in names:
        so = out / f"libhowl_{name}.so"
        vobjs = []
        r = None
        for f in ("res8", "frontend"):   # the two files that carry HOWL_DIAG_* switches
            obj = out / f"{f}_{name}.o"
            r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", *VARIANTS[name],
                                "-c", str(CSRC / f"{f}.hip"), "-o", str(obj)], capture_output=True, text=True)
            if r.returncode != 0:
                break
            vobjs.append(str(obj))
        if r.returncode == 0:
            r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", *vobjs, *objs, "-o", str(so)],
                               capture_output=True, text=True)
        if r.returncode != 0:
            print(f"{name}: build failed\n{r.stderr[-2000:]}")
            continue
        env = dict(os.environ, HOWL_HIP_LIBRARY=str(so))
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
        print(f"{name:14s} {r.stdout.strip() or r.stderr[-800:]}", flush=True)


if __name__ == "__main__":
    main()
