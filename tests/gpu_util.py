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
