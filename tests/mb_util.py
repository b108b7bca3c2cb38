"""Shared pieces of the MobileNetClassifier parity tests."""
import numpy as np
import torch

from oracle import mobilenet as omb


def oracle_step(sd, x, labels, keep):
    """Training-mode forward + parameter gradients of the mean cross-entropy; ``sd`` is not modified."""
    names = omb.mobilenet_param_names()
    osd = {k: v.clone() for k, v in sd.items()}
    params = [osd[n].requires_grad_(True) for n in names]
    logits = omb.mobilenet_forward(osd, x, True, keep)
    grads = torch.autograd.grad(torch.nn.functional.cross_entropy(logits, labels), params)
    return logits.detach(), grads, {k: v.detach() for k, v in osd.items()}


def rel_l2(a, b):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a)).double()
    b = b.detach().cpu().double()
    return ((a - b).norm() / b.norm()).item()


def check_grads(got, sd, x, labels, keep, ref_grads, eps=1e-6):
    """Per parameter tensor: relative L2 distance to the oracle's gradient.

    53 BatchNorm+ReLU6 layers make the fp32 gradient of this network discontinuous in its inputs at the 1e-2 level (an
    activation within rounding of 0 or 6 flips its mask): the oracle evaluated on an input perturbed by 1e-6 of white noise
    moves by that much.  The bound is therefore calibrated per tensor on that self-sensitivity (x4, two noise draws), with a
    floor of 1e-2: at batch 6 a channel of the last layers has 6 values, ONE of them within rounding of 0 or 6 (where the
    kernels' y = z * scale + shift and the oracle's ((z - mean) * rstd) * gamma + beta may fall on different sides) moves that
    channel's gradient by 1/6 and every tensor below it by ~5e-3 -- seen with exactly one of 1280 channels off and the other
    1279 equal to 4e-9.  A wrong kernel is off by O(1).  Gradients that are zero by construction (a bias in front of a BatchNorm, the
    BatchNorm bias of a projection feeding conv+BatchNorm) must be rounding noise.  ``eps`` is the relative size of
    the perturbation; pass the size of the actual input difference when the inputs themselves are only equal to a
    tolerance (features from the HIP frontend vs the oracle frontend)."""
    names = omb.mobilenet_param_names()
    perts = []
    for seed in (7, 8):   # additive white noise: BatchNorm cancels a uniform rescaling
        noise = torch.randn(x.shape, generator=torch.Generator().manual_seed(seed))
        perts.append(oracle_step(sd, x + eps * noise, labels, keep)[1])
    scale = max(r.abs().max().item() for r in ref_grads)
    worst = 0.0
    for i, (n, g, r) in enumerate(zip(names, got, ref_grads)):
        if r.norm().item() < 1e-4 * scale:
            assert float(torch.as_tensor(np.asarray(g.detach().cpu() if torch.is_tensor(g) else g)).abs().max()) < 1e-4 * scale, n
            continue
        err, sens = rel_l2(g, r), max(rel_l2(p[i], r) for p in perts)
        assert err < max(1e-2, 4.0 * sens), f"grad {n}: relative L2 error {err:.2e} (oracle self-sensitivity {sens:.2e})"
        worst = max(worst, err)
    return worst
