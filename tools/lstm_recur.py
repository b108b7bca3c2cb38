"""Average duration of the two LSTM recurrence launches (HIP events inside the library) in the seq-lstm training step:
python tools/lstm_recur.py [B ...]   (38 of 51 frames run, as BASELINE configs[3])"""
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
from howl_amd.training.fused import FusedTrainer  # noqa: E402
from howl_amd.utils.synth import synthetic_pcm  # noqa: E402

dev = torch.device("cuda:0")
lb = hlib.get()
for B in [int(a) for a in sys.argv[1:]] or [512]:
    pcm = synthetic_pcm(B, 8000).to(dev)
    std = StandardAudioTransform().to(dev).eval()
    zmuv = ZmuvTransform().to(dev)
    zmuv.update(std(pcm[:8]))
    steps = 38
    lengths = torch.full((B,), steps)
    targets = torch.tensor([[0, 1, 2]] * B).to(dev)
    tl = torch.tensor([3] * B)
    model = RegisteredModel.find_registered_class("seq-lstm")(5).to(dev).train()
    tr = FusedTrainer(model, std, zmuv, lr=1e-4, weight_decay=1e-5)
    for _ in range(5):
        tr.step_sequence(pcm, lengths, targets, tl, 4, max_target=3)
    torch.cuda.synchronize()
    lb.call("howl_profile_enable", 1)
    for _ in range(30):
        tr.step_sequence(pcm, lengths, targets, tl, 4, max_target=3)
    torch.cuda.synchronize()
    lb.call("howl_profile_enable", 0)
    out = []
    for tag, reset in (("lstm_fwd", 0), ("lstm_bwd", 1)):       # a read with reset clears every tag
        tot, cnt, work = ctypes.c_double(0), ctypes.c_int(0), ctypes.c_double(0)
        lb.call("howl_profile_read_work", tag.encode(), ctypes.byref(tot), ctypes.byref(cnt), ctypes.byref(work), reset)
        us = tot.value / max(cnt.value, 1) * 1e3
        out.append(f"{tag} {us:.1f} us = {us / steps:.2f} us/step")
    print(f"B={B}: " + ", ".join(out), flush=True)
