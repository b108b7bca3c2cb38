"""seq-lstm CTC training step (BASELINE configs[3]: batch 512, 0.5 s), for profiling:  python tools/lstm_step.py [steps] [fused|autograd]"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("NUM_MELS", "40")
import torch  # noqa: E402

from howl_amd import ops  # noqa: E402
from howl_amd.data.transform.operator import ZmuvTransform  # noqa: E402
from howl_amd.data.transform.transform import StandardAudioTransform  # noqa: E402
from howl_amd.model import RegisteredModel  # noqa: E402
from howl_amd.utils.synth import synthetic_pcm  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
B, L, C = 512, 8000, 5
pcm = synthetic_pcm(B, L).to(dev)
std = StandardAudioTransform().to(dev).eval()
zmuv = ZmuvTransform().to(dev)
zmuv.update(std(pcm[:8]))
from howl_amd.training.fused import FusedTrainer  # noqa: E402

lengths = torch.full((B,), 38)
targets = torch.tensor([[0, 1, 2]] * B).to(dev)
tl = torch.tensor([3] * B)
mode = sys.argv[2] if len(sys.argv) > 2 else "fused"
model = RegisteredModel.find_registered_class("seq-lstm")(C).to(dev).train()
if mode == "fused":      # training.fused.FusedTrainer.step_sequence: explicit launches, flat AdamW
    tr = FusedTrainer(model, std, zmuv, lr=1e-4, weight_decay=1e-5)

    def step():
        return tr.step_sequence(pcm, lengths, targets, tl, 4, max_target=3)
else:                    # autograd path + torch.optim.AdamW, as in the reference's loop
    opt = torch.optim.AdamW(model.parameters(), 1e-4, weight_decay=1e-5)

    def step():
        feats = std.log_mel_for_model(pcm, zmuv)
        sc = model(feats, lengths)
        loss = ops.ctc_loss(sc, targets, lengths, tl, 4)       # fused log_softmax + CTCLoss(blank=4)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"seq-lstm B={B} ({mode}): {dt * 1e3:.3f} ms/step {B / dt:.0f} utt/s loss {loss.item():.4f}")
