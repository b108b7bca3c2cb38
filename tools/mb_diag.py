"""Gradient-parity statistics of MobileNetClassifier against the oracle (per-tensor relative L2), on the GPU box."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
os.environ.setdefault("NUM_MELS", "40")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import mobilenet as omb  # noqa: E402
from test_gpu_mobilenet import make_mobilenet, rel_l2  # noqa: E402

DEV = torch.device("cuda:0")
for B, T, C in ((8, 81, 12), (16, 41, 4), (32, 81, 12)):
    torch.manual_seed(B + T)
    x = torch.randn(B, 3, 40, T) * 1.5
    labels = torch.arange(B) % C
    model, sd = make_mobilenet(C)
    model.dropout_p = 0.0
    logits = model(x.to(DEV), None)
    torch.nn.functional.cross_entropy(logits, labels.to(DEV)).backward()
    names = omb.mobilenet_param_names()
    osd = {k: v.clone() for k, v in sd.items()}
    params = [osd[n].requires_grad_(True) for n in names]
    ref = omb.mobilenet_forward(osd, x, True, None)
    ref_grads = torch.autograd.grad(torch.nn.functional.cross_entropy(ref, labels), params)
    # sensitivity of the oracle itself: the same step with the input perturbed by one part in 1e6
    osd2 = {k: v.clone() for k, v in sd.items()}
    p2 = [osd2[n].requires_grad_(True) for n in names]
    ref2 = omb.mobilenet_forward(osd2, x * (1 + 1e-6), True, None)
    g2 = torch.autograd.grad(torch.nn.functional.cross_entropy(ref2, labels), p2)
    scale = max(r.abs().max().item() for r in ref_grads)
    rows = []
    for n, p, r, r2 in zip(names, model.hot_parameters(), ref_grads, g2):
        if r.norm().item() >= 1e-4 * scale:
            rows.append((rel_l2(p.grad, r), rel_l2(r2, r), n))
    e = np.array([q[0] for q in rows])
    s = np.array([q[1] for q in rows])
    print(f"B={B} T={T}: logits err {(logits.detach().cpu() - ref.detach()).abs().max().item():.2e}; grad rel-L2 hip-vs-oracle "
          f"median {np.median(e):.2e} max {e.max():.2e} | oracle self-sensitivity (1e-6 input perturbation) median "
          f"{np.median(s):.2e} max {s.max():.2e}")
    print("   worst:", [(f"{a:.1e}", f"{b:.1e}", n) for a, b, n in sorted(rows, reverse=True)[:3]], flush=True)
