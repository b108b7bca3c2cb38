"""Diagnostic: one fused training step (features already on the device) launched eagerly vs replayed as a captured hipGraph.
python tools/graph_probe.py"""
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

dev = torch.device("cuda:0")
B, C = 512, 12
pcm = synthetic_pcm(B, 16000).to(dev)
labels = (torch.arange(B) % C).to(dev)
std = StandardAudioTransform().to(dev).eval()
zmuv = ZmuvTransform().to(dev)
zmuv.update(std(pcm[:8]))
for name, lr in (("res8", 0.01), ("mobilenet", 0.001)):
    model = RegisteredModel.find_registered_class(name)(C).to(dev).train()
    if name == "mobilenet":
        model.dropout_p = 0.0
    tr = FusedTrainer(model, std, zmuv, lr=lr)
    feat = tr.features(pcm)

    def step():
        return tr.step_on_features(feat, labels)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 50 * 1e3
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(graph):
            step()
        for _ in range(5):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            graph.replay()
        torch.cuda.synchronize()
        replay = (time.perf_counter() - t0) / 50 * 1e3
        print(f"{name}: eager {eager:.4f} ms/step (features resident), graph replay {replay:.4f} ms/step")
    except Exception as e:  # noqa: BLE001
        print(f"{name}: eager {eager:.4f} ms/step; capture failed: {type(e).__name__}: {str(e)[:300]}")
