"""Shared helpers for the -m gpu parity tests (HIP path vs oracle / golden vectors)."""
import numpy as np
import torch

from oracle import frontend as ofe
from oracle import models as om

DEV = torch.device("cuda:0")


def t(a):
    return torch.from_numpy(np.asarray(a))


def maxerr(a, b):
    a = a.detach().cpu().double() if torch.is_tensor(a) else torch.as_tensor(a).double()
    b = b.detach().cpu().double() if torch.is_tensor(b) else torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return (a - b).abs().max().item() if a.numel() else 0.0


def golden_features(golden):
    """ZMUV-normalised (B,3,40,81) features of the six GSC clips, from the reference's own outputs."""
    g2, g4 = golden("g2_frontend_gsc"), golden("g4_zmuv")
    z = ofe.Zmuv()
    z.mean, z.mean2 = t(g4["mean"]), t(g4["mean2"])
    return z(t(g2["feats"])), z


def make_res8(C, state=None, train=True):
    from howl_amd.model import RegisteredModel
    model = RegisteredModel.find_registered_class("res8")(C)
    sd = state or om.res8_init(C)
    model.load_state_dict({k: v.clone() for k, v in sd.items()})
    model = model.to(DEV)
    return model.train() if train else model.eval()


def res8_relu_masks(model, B, T, M):
    """The ReLU on/off decisions of the res8 kernels' last training forward, in the oracle's layouts (seven bool tensors:
    (B, 45, T, M) for conv0 -- rows beyond 3 (T // 3) are dropped by the pooling and reported off --, (B, 45, T/3, M/4) for the
    3x3 layers), read from what the forward saves for the backward (include/howl_hip.h HowlRes8Saved): conv0's 12-bit
    patterns per pooled cell, the sign of s_i for the layers without a residual add, its sign BIT for those with one."""
    buf = next(v for k, v in model._buffers_cache.items() if k[0] == B and k[1] == T and k[2] == M)
    H, W = T // 3, M // 4
    ns = W // 10                                  # column strips of 10 pooled columns (80 mel bins: two)
    nr = 1 if H <= 27 else -(-H // 27)            # row strips beyond 83 frames, Hs rows each, the last one padded
    Hs = -(-H // nr)

    def unstrip(t):      # block (b * nr + r) * ns + c of (45, Hs, 10) = rows r Hs .., columns 10 c .. of utterance b
        t = t.detach().cpu().reshape(B, nr, ns, 45, Hs, 10)
        return t.permute(0, 3, 1, 4, 2, 5).reshape(B, 45, nr * Hs, W)[:, :, :H]

    bits = unstrip(buf.mask0).to(torch.int32) & 0xFFFF
    m0 = torch.zeros(B, 45, T, M, dtype=torch.bool)
    for tl in range(3):
        for fl in range(4):
            m0[:, :, tl:3 * H:3, fl::4] = ((bits >> (4 * tl + fl)) & 1).bool()
    masks = [m0]
    for i in range(1, 7):
        s_i = unstrip(buf.s[i])
        masks.append(torch.signbit(s_i) if i % 2 == 0 else s_i > 0)
    return masks


def res8_oracle_with_kernel_relus(model, x, labels, B, T, M, C, flip_tol=3e-6):
    """The oracle's training step on features x with the kernels' own ReLU decisions: where the two disagree the oracle's
    pre-activation must lie within rounding of zero (asserted: |z| < flip_tol, a handful of elements), and with the decisions
    shared the gradients have to agree to summation-order rounding whatever the batch size.  Returns (logits, grads by name,
    number of flipped decisions)."""
    names = om.res8_param_names()
    pre = []
    om.res8_forward(om.res8_init(C), x.contiguous(), True, pre_relu=pre)
    masks = res8_relu_masks(model, B, T, M)
    flips = 0
    H = T // 3
    for i, (z, m) in enumerate(zip(pre, masks)):
        if i == 0:
            z, m = z[:, :, :3 * H], m[:, :, :3 * H]
        bad = (z > 0) != m
        flips += int(bad.sum())
        if bad.any():
            assert z[bad].abs().max().item() < flip_tol, (i, int(bad.sum()), z[bad].abs().max().item())
    masks[0][:, :, 3 * H:] = pre[0][:, :, 3 * H:] > 0      # rows the pooling drops: no gradient flows there
    assert flips <= 2 + 2e-5 * sum(z.numel() for z in pre), flips
    sd = om.res8_init(C)
    params = [sd[n].requires_grad_(True) for n in names]
    ref = om.res8_forward(sd, x.contiguous(), True, relu_masks=masks)
    grads = torch.autograd.grad(torch.nn.functional.cross_entropy(ref, labels), params)
    return ref.detach(), dict(zip(names, grads)), flips, sd


def res8_plain_oracle_grads(x, labels, C):
    """The oracle's training step with its OWN ReLU decisions (no mask plumbing): at full batch sizes one flipped decision moves
    a gradient by ~1e-6 relative, so the kernels must also agree with this at 5e-5 (the shared-decision
    comparison must never be the only thing between a strip bug and green)."""
    names = om.res8_param_names()
    sd = om.res8_init(C)
    params = [sd[n].requires_grad_(True) for n in names]
    ref = om.res8_forward(sd, x.contiguous(), True)
    grads = torch.autograd.grad(torch.nn.functional.cross_entropy(ref, labels), params)
    return ref.detach(), dict(zip(names, grads))
