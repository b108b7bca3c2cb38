# NOTE (round 5): the HOWL_DIAG_* branches this tool compiles were removed from the product kernels (tools/strip_diag.py);
# it builds against the sources of commit 37e3835 (`git worktree add /tmp/howl_r4 37e3835` and run it there).
"""Per-phase timelines (one s_memtime stamp per wave and phase; diagnostic build with -DHOWL_DIAG_PROBE) of the round-4 3x3
kernels at 512 x 1 s: the forward convolution (last layer's launch, workgroup 0) and both roles of the backward pair (layer
1's launch: block 0 = data gradient, block 8 = weight gradient).
    python tools/probe_step4.py build      (here: cross-compiles build/probe/libhowl_probe.so)
    python tools/probe_step4.py run        (GPU box)"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "howl_amd" / "csrc"
OUT = ROOT / "build" / "probe"
SO = OUT / "libhowl_probe.so"

CHILD = r"""
import ctypes, os, sys
sys.path.insert(0, %r)
os.environ.setdefault("NUM_MELS", "40")
import torch
from howl_amd import lib as hlib
from howl_amd.data.transform.operator import ZmuvTransform
from howl_amd.data.transform.transform import StandardAudioTransform
from howl_amd.model import RegisteredModel
from howl_amd.training.fused import FusedRes8Trainer
from howl_amd.utils.synth import res8_closed_form_state, synthetic_pcm
B = 512; dev = torch.device("cuda:0"); lb = hlib.get()
pcm = synthetic_pcm(B, 16000).to(dev); labels = (torch.arange(B) %% 12).to(dev)
std = StandardAudioTransform().to(dev).eval(); zmuv = ZmuvTransform().to(dev); zmuv.update(std(pcm[:8]))
model = RegisteredModel.find_registered_class("res8")(12).to(dev)
model.load_state_dict(res8_closed_form_state(12), strict=False); model.train()
tr = FusedRes8Trainer(model, std, zmuv, lr=0.0)
feat = tr.features(pcm)
for _ in range(3): tr.step_on_features(feat, labels)
lb.cdll.howl_diag_set_probe.argtypes = [ctypes.c_void_p, ctypes.c_int]
FWD = "entry requested folded zeroed weights setup-bar tile0 | per utterance: A bar B bar epi | done".split()
def dump(title, block, fn, names):
    buf = torch.zeros(12 * 64, dtype=torch.int64, device=dev)
    assert lb.cdll.howl_diag_set_probe(buf.data_ptr(), block) == 0
    fn(); torch.cuda.synchronize()
    assert lb.cdll.howl_diag_set_probe(None, 0) == 0
    t = buf.cpu().view(12, 64)
    print("stamps written:", int((t != 0).sum()), "of", t.numel(), flush=True)
    if not (t != 0).any():
        return
    t0 = int(t[t != 0].min())
    print("==", title, "(ticks ~ shader cycles; first column = entry relative to the first wave, then deltas)")
    print("   ", names)
    for w in (0, 4, 8, 3, 7, 11):
        row = [int(v) - t0 for v in t[w] if int(v) != 0]
        d = [row[0]] + [row[i] - row[i - 1] for i in range(1, len(row))]
        print("wave %%2d:" %% w, " ".join("%%6d" %% v for v in d), "| total", row[-1])
dump("forward conv, layer 6, workgroup 0", 0, lambda: model._launch_forward(feat),
     "entry requested folded zeroed weights setup-barrier tile0 | per utterance: phaseA barrier phaseB barrier epilogue | done")
dump("pair, data-gradient role (block 0), layer 1", 0, lambda: tr.step_on_features(feat, labels),
     "entry requested folded zeroed weights setup-barrier tile0 | per utterance: phaseA barrier phaseB barrier epilogue | done")
dump("pair, weight-gradient role (block 8), layer 1", 8, lambda: tr.step_on_features(feat, labels),
     "entry prologue | per utterance: phase1 barrier phase2 barrier | partials")
""" % str(ROOT)


def build():
    OUT.mkdir(parents=True, exist_ok=True)
    obj = OUT / "res8_probe.o"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-DHOWL_DIAG_PROBE",
                    "-c", str(CSRC / "res8.hip"), "-o", str(obj)], check=True)
    others = [str(p) for p in sorted((ROOT / "build" / "obj").glob("*.o")) if p.name != "res8.o"]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", str(obj), *others, "-o", str(SO)], check=True)
    print("built", SO)


def run(so=SO):
    env = dict(os.environ, HOWL_HIP_LIBRARY=str(so))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    print(r.stdout.strip(), flush=True)
    print(r.stderr[-2500:], flush=True)


if __name__ == "__main__":
    if sys.argv[1:] == ["build"]:
        build()
    else:
        run(*[Path(a).resolve() for a in sys.argv[2:3]])
