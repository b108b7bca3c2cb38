"""Longer sanity runs on the GPU box: a few hundred fused steps of each trainable model on separable synthetic tones; the loss
must fall, every parameter must stay finite, and a repeat from the same state must be bit-identical."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("NUM_MELS", "40")
import torch  # noqa: E402

from howl_amd.data.transform.operator import ZmuvTransform  # noqa: E402
from howl_amd.data.transform.transform import StandardAudioTransform  # noqa: E402
from howl_amd.model import RegisteredModel  # noqa: E402
from howl_amd.training.fused import FusedTrainer  # noqa: E402
from howl_amd.utils.synth import synthetic_pcm  # noqa: E402

from howl_amd.settings import SETTINGS  # noqa: E402

dev = torch.device("cuda:0")
# (model, batch, labels, lr, steps, mel bins, samples): the last res8 run is the reference's STOCK 80 mel bins on 2-s windows -- every
# utterance two column strips x two row strips in the kernels
for name, B, C, lr, steps, mels, L in (("res8", 256, 12, 0.01, 300, 40, 16000), ("mobilenet", 128, 12, 0.001, 150, 40, 16000),
                                       ("mobilenet", 512, 12, 0.001, 120, 40, 16000), ("res8", 128, 12, 0.01, 300, 80, 32000)):
    finals = []
    SETTINGS.audio_transform.num_mels = mels
    for rep in range(2):
        torch.manual_seed(0)
        pcm = synthetic_pcm(B, L).to(dev)                  # tone frequency depends on b mod 64: labels are learnable
        labels = (torch.arange(B) % C).to(dev)
        std = StandardAudioTransform().to(dev).eval()
        zmuv = ZmuvTransform().to(dev)
        zmuv.update(std(pcm[:8]))
        model = RegisteredModel.find_registered_class(name)(C).to(dev).train()
        if name == "mobilenet":
            model.dropout_p = 0.0
        tr = FusedTrainer(model, std, zmuv, lr=lr)
        losses = []
        for i in range(steps):
            loss = tr.step(pcm, labels)
            if i % 25 == 0 or i == steps - 1:
                losses.append(round(loss.item(), 4))
        assert all(torch.isfinite(p).all() for p in model.parameters()), name
        finals.append(tr.fp.flat.clone())
        if rep == 0:
            model.eval()
            with torch.no_grad():
                acc = (model(tr.features(pcm), None).argmax(1) == labels).float().mean().item()
            print(f"{name} ({mels} mel bins, {L / 16000:g} s): loss {losses[0]} -> {losses[-1]} over {steps} steps (every 25th: {losses}); train-set accuracy in eval "
                  f"mode {acc:.3f}", flush=True)
            assert losses[-1] < 0.5 * losses[0], (name, losses)
    print(f"{name}: repeat run bit-identical: {torch.equal(finals[0], finals[1])}", flush=True)
    assert torch.equal(finals[0], finals[1])
SETTINGS.audio_transform.num_mels = 40
# sequence objective: seq-lstm + fused log_softmax/CTC on 0.5 s tones whose frequency class picks one of four label sequences
finals = []
for rep in range(2):
    torch.manual_seed(0)
    B, C, blank = 256, 5, 4
    pcm = synthetic_pcm(B, 8000).to(dev)
    seqs = [[0, 1, 2], [2, 1], [3], [1, 1, 0]]                 # incl. a repeated label and different lengths
    targets = torch.zeros(B, 3, dtype=torch.long)
    tl = torch.zeros(B, dtype=torch.long)
    for b in range(B):
        q = seqs[b % 4]
        targets[b, : len(q)] = torch.tensor(q)
        tl[b] = len(q)
    std = StandardAudioTransform().to(dev).eval()
    zmuv = ZmuvTransform().to(dev)
    zmuv.update(std(pcm[:8]))
    model = RegisteredModel.find_registered_class("seq-lstm")(C).to(dev).train()
    tr = FusedTrainer(model, std, zmuv, lr=0.003)
    lengths = torch.full((B,), 38)
    losses = []
    for i in range(200):
        loss = tr.step_sequence(pcm, lengths, targets, tl, blank, max_target=3)
        if i % 25 == 0 or i == 199:
            losses.append(round(loss.item(), 4))
    assert all(torch.isfinite(p).all() for p in model.parameters())
    finals.append(tr.fp.flat.clone())
    if rep == 0:
        print(f"seq-lstm/ctc: loss {losses[0]} -> {losses[-1]} over 200 steps (every 25th: {losses})", flush=True)
        assert losses[-1] < 0.5 * losses[0], losses
print(f"seq-lstm/ctc: repeat run bit-identical: {torch.equal(finals[0], finals[1])}", flush=True)
assert torch.equal(finals[0], finals[1])
print("soak ok")
