"""Throughput of the other BASELINE.json configurations (parity-test cases, not the bench line): res8 at configs[0]/[1]
geometry and the seq-lstm CTC step of configs[3].  Prints one line per configuration."""
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
from howl_amd.training.fused import FusedRes8Trainer, FusedTrainer  # noqa: E402
from howl_amd.utils.synth import res8_closed_form_state, synthetic_pcm  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, warmup=5, steps=30):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


for name, B, L, C in (("C1 res8 B=64 1s C=30", 64, 16000, 30), ("C2 res8 B=256 0.5s C=4", 256, 8000, 4),
                      ("C3/gpu res8 B=512 1s C=12", 512, 16000, 12)):
    pcm = synthetic_pcm(B, L).to(dev)
    labels = (torch.arange(B) % C).to(dev)
    std = StandardAudioTransform().to(dev).eval()
    zmuv = ZmuvTransform().to(dev)
    zmuv.update(std(pcm[:8]))
    model = RegisteredModel.find_registered_class("res8")(C).to(dev)
    model.load_state_dict(res8_closed_form_state(C), strict=False)
    model.train()
    tr = FusedRes8Trainer(model, std, zmuv, lr=0.01)
    dt = timeit(lambda: tr.step(pcm, labels))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            tr.step(pcm, labels)
    torch.cuda.current_stream().wait_stream(s)
    msg = ""
    try:
        step0 = tr.step_count
        with torch.cuda.graph(g):
            tr.step(pcm, labels)
        dtg = timeit(lambda: g.replay())
        msg = f" | hipGraph replay {dtg * 1e3:.3f} ms {B / dtg:.0f} utt/s (fixed AdamW step index {step0 + 1})"
    except Exception as e:  # noqa: BLE001
        msg = f" | graph capture failed: {type(e).__name__}: {e}"
    print(f"{name}: {dt * 1e3:.3f} ms/step {B / dt:.0f} utt/s{msg}", flush=True)

# configs[3]: seq-lstm, CTC, batch 512, 0.5 s
B, L, C = 512, 8000, 5
pcm = synthetic_pcm(B, L).to(dev)
std = StandardAudioTransform().to(dev).eval()
zmuv = ZmuvTransform().to(dev)
zmuv.update(std(pcm[:8]))
model = RegisteredModel.find_registered_class("seq-lstm")(C).to(dev).train()
lengths = torch.full((B,), 38)
targets = torch.tensor([[0, 1, 2]] * B).to(dev)
tl = torch.tensor([3] * B)
seq_trainer = FusedTrainer(model, std, zmuv, lr=1e-4, weight_decay=1e-5)


def lstm_step():      # frontend -> LSTM + head -> fused log_softmax + CTC(blank=4) -> backward -> flat AdamW
    seq_trainer.step_sequence(pcm, lengths, targets, tl, 4, max_target=3)


dt = timeit(lstm_step, steps=20)
print(f"C4 seq-lstm B=512 0.5s CTC: {dt * 1e3:.3f} ms/step {B / dt:.0f} utt/s", flush=True)

# configs[4]: mobilenet GSC-12, 512 x 1 s per GPU (2048 over 4 GPUs), noise-augment collate on device in front of the step
B, L, C = 512, 16000, 12
pcm = synthetic_pcm(B, L).to(dev)
labels = (torch.arange(B) % C).to(dev)
std = StandardAudioTransform().to(dev).eval()
zmuv = ZmuvTransform().to(dev)
zmuv.update(std(pcm[:8]))
model = RegisteredModel.find_registered_class("mobilenet")(C).to(dev).train()
tr = FusedRes8Trainer(model, std, zmuv, lr=0.001)   # envs/mobilenet.env: LEARNING_RATE=0.001, WEIGHT_DECAY=0
from howl_amd.data.collate import DeviceCollate  # noqa: E402

bank_lengths = torch.full((B,), L, dtype=torch.long)
collate = DeviceCollate(pcm, bank_lengths, labels, max_len=L, seed=0)   # timeshift + white / salt-pepper noise on device
ids = list(range(B))


def mb_step():
    batch = collate(ids)
    audio = batch.audio_data
    if audio.shape[1] != L:   # timeshift crops; the step geometry is fixed at 1 s -> right-pad like batchify
        audio = torch.nn.functional.pad(audio, (0, L - audio.shape[1]))
    tr.step(audio, batch.labels)


dt = timeit(mb_step, warmup=3, steps=10)
dt0 = timeit(lambda: tr.step(pcm, labels), warmup=2, steps=10)
print(f"C5 mobilenet B=512 1s C=12: {dt * 1e3:.3f} ms/step {B / dt:.0f} utt/s with the device collate/augment in the loop "
      f"({dt0 * 1e3:.3f} ms, {B / dt0:.0f} utt/s without)", flush=True)
