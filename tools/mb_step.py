"""MobileNet (BASELINE configs[4] per-GPU share: 512 x 1 s, 12 labels) fused training step, for profiling:
    python tools/mb_step.py [steps] [batch]"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("NUM_MELS", "40")
import torch  # noqa: E402

from howl_amd.data.transform.operator import ZmuvTransform  # noqa: E402
from howl_amd.data.transform.transform import StandardAudioTransform  # noqa: E402
from howl_amd.model import RegisteredModel  # noqa: E402
from howl_amd.training.fused import FusedTrainer  # noqa: E402
from howl_amd.utils.synth import synthetic_pcm  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
pcm = synthetic_pcm(B, 16000).to(dev)
labels = (torch.arange(B) % 12).to(dev)
std = StandardAudioTransform().to(dev).eval()
zmuv = ZmuvTransform().to(dev)
zmuv.update(std(pcm[:8]))
model = RegisteredModel.find_registered_class("mobilenet")(12).to(dev).train()
tr = FusedTrainer(model, std, zmuv, lr=0.001)
for _ in range(3):
    tr.step(pcm, labels)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = tr.step(pcm, labels)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"mobilenet B={B}: {dt * 1e3:.3f} ms/step {B / dt:.0f} utt/s loss {loss.item():.4f}")
