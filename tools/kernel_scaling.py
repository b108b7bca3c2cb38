"""Per-kernel HIP-event timings of the res8 step at several batch sizes (fixed-overhead vs per-utterance cost)."""
import ctypes
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("NUM_MELS", "40")
import torch  # noqa: E402

from howl_amd import lib as hlib  # noqa: E402
from howl_amd.data.transform.operator import ZmuvTransform  # noqa: E402
from howl_amd.data.transform.transform import StandardAudioTransform  # noqa: E402
from howl_amd.model import RegisteredModel  # noqa: E402
from howl_amd.training.fused import FusedRes8Trainer  # noqa: E402
from howl_amd.utils.synth import res8_closed_form_state, synthetic_pcm  # noqa: E402

dev = torch.device("cuda:0")
lb = hlib.get()
L = int(float(sys.argv[1]) * 16000) if len(sys.argv) > 1 else 16000
for B in (256, 512, 1024, 2048):
    pcm = synthetic_pcm(B, L).to(dev)
    labels = (torch.arange(B) % 12).to(dev)
    std = StandardAudioTransform().to(dev).eval()
    zmuv = ZmuvTransform().to(dev)
    zmuv.update(std(pcm[:8]))
    model = RegisteredModel.find_registered_class("res8")(12).to(dev)
    model.load_state_dict(res8_closed_form_state(12), strict=False)
    model.train()
    tr = FusedRes8Trainer(model, std, zmuv, lr=0.01)
    for _ in range(5):
        tr.step(pcm, labels)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(20):
        tr.step(pcm, labels)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    lb.call("howl_profile_enable", 1)
    for _ in range(10):
        tr.step(pcm, labels)
    torch.cuda.synchronize()
    lb.call("howl_profile_enable", 0)
    row = [f"B={B:5d} L={L} step={dt * 1e3:7.3f} ms {B / dt:9.0f} utt/s |"]
    tags = ["logmel", "conv3x3_fwd", "conv3x3_dgrad", "wgrad"]
    for i, tag in enumerate(tags):
        tot, cnt = ctypes.c_double(0), ctypes.c_int(0)
        lb.call("howl_profile_read", tag.encode(), ctypes.byref(tot), ctypes.byref(cnt), int(i == len(tags) - 1))
        row.append(f"{tag} {tot.value / max(cnt.value, 1) * 1e3:7.1f} us")
    print(" ".join(row), flush=True)
