"""MobileNetClassifier kernels (howl_amd/csrc/mobilenet.hip) on the hipemu CPU emulator vs the oracle restatement
(oracle/mobilenet.py; parity unpinned against torchvision -- see its header): layer table, forward in training and
eval mode, BatchNorm buffer updates, every parameter gradient."""
import ctypes

import numpy as np
import pytest
import torch

from emu_util import emu_lib, ptr
from howl_amd.lib import HowlMbLayer
from mb_util import check_grads, oracle_step
from oracle import mobilenet as om


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


def layer_table(lib):
    out = []
    for i in range(lib.cdll.howl_mobilenet_num_layers()):
        d = HowlMbLayer()
        lib.call("howl_mobilenet_layer", i, ctypes.byref(d))
        out.append(d)
    return out


def flat_params(lib, sd, C):
    names = om.mobilenet_param_names()
    flat = np.concatenate([sd[n].numpy().reshape(-1) for n in names]).astype(np.float32)
    assert flat.size == lib.cdll.howl_mobilenet_param_floats(C)
    return flat, names


def flat_buffers(sd):
    out = []
    for l in om.layer_table():
        out += [sd[l["bn"] + ".running_mean"].numpy(), sd[l["bn"] + ".running_var"].numpy()]
    return np.concatenate(out).astype(np.float32)


def test_layer_table_matches_oracle(lib):
    tab, otab = layer_table(lib), om.layer_table()
    assert len(tab) == len(otab) == 53
    kinds = {"dense": 0, "pw": 1, "dw": 2}
    acts = {"none": 0, "relu6": 1, "relu": 2}
    off = 0
    for d, o in zip(tab, otab):
        assert (d.kind, d.cin, d.cout, d.stride, d.pad_h, d.pad_w, d.act, d.bias, d.pool) == \
               (kinds[o["kind"]], o["cin"], o["cout"], o["stride"], o["pad"][0], o["pad"][1], acts[o["act"]], int(o["bias"]),
                int(o["pool"]))
        assert (d.res_src >= 0) == o["res"]
        # state-dict key scheme documented in include/howl_hip.h
        if d.feat < 0:
            key = "downsample.0"
        elif d.sub < 0:
            key = f"model.features.{d.feat}.0"
        elif d.wrapped:
            key = f"model.features.{d.feat}.conv.{d.sub}.0"
        else:
            key = f"model.features.{d.feat}.conv.{d.sub}"
        assert key == o["key"]
        assert d.w_off == off
        off += int(np.prod(om.conv_shape(o))) + (o["cout"] if o["bias"] else 0) + 2 * o["cout"]
    assert lib.cdll.howl_mobilenet_buffer_floats() == sum(2 * o["cout"] for o in otab)
    # known answer from the literature: torchvision's mobilenet_v2 (width 1.0) has 3,504,872 parameters with its 1000-way
    # classifier, 2,223,872 of them in `features` -- the layer table (+ the 36-parameter downsample stem in front) must add up to
    # exactly that.  (torchvision itself is not available here: this pins the architecture, not the arithmetic.)
    downsample = 3 * 1 * 9 + 3 + 3 + 3
    assert lib.cdll.howl_mobilenet_param_floats(1000) - downsample == 3504872
    assert lib.cdll.howl_mobilenet_param_floats(1000) - downsample - (1000 * 1280 + 1000) == 2223872


@pytest.mark.parametrize("B,T,dropout", [(6, 41, False), (5, 30, True)])
def test_forward_backward_vs_oracle(lib, B, T, dropout):
    C, M = 5, 40
    torch.manual_seed(B * 100 + T)
    x = torch.randn(B, 3, M, T) * 1.5                       # the kernels read channel 0 through strides
    labels = torch.arange(B) % C
    keep = (torch.rand(B, om.LAST_CHANNEL) >= 0.2).float() if dropout else None
    sd = om.mobilenet_init(C)
    flat, names = flat_params(lib, sd, C)
    bufs = flat_buffers(sd)
    xn = np.ascontiguousarray(x.numpy())
    sb, sm, st = 3 * M * T, T, 1
    ws = np.zeros(lib.cdll.howl_mobilenet_workspace_bytes(B, M, T, C), np.uint8)
    logits = np.zeros((B, C), np.float32)
    mask = None if keep is None else np.ascontiguousarray(keep.numpy())
    scale = 1.0 / (1.0 - om.DROPOUT_P) if dropout else 1.0
    lib.call("howl_mobilenet_fwd", ptr(flat), ptr(bufs), C, ptr(xn), sb, sm, st, B, M, T, 1, ptr(mask), scale, ptr(logits),
             ptr(ws), ws.size, None)

    ref, grads, osd = oracle_step(sd, x, labels, keep)
    np.testing.assert_allclose(logits, ref.numpy(), rtol=0, atol=5e-4)   # within the 1e-3 of BASELINE's north star
    assert (logits.argmax(1) == ref.numpy().argmax(1)).all()
    np.testing.assert_allclose(bufs, flat_buffers({k: v.detach() for k, v in osd.items()}), rtol=1e-4, atol=1e-5)

    p = torch.softmax(ref, 1)
    p[torch.arange(B), labels] -= 1
    dlogits = np.ascontiguousarray((p / B).numpy().astype(np.float32))
    g = np.full(flat.size, np.nan, np.float32)
    lib.call("howl_mobilenet_bwd", ptr(flat), C, ptr(xn), sb, sm, st, B, M, T, ptr(mask), scale, ptr(dlogits), ptr(g), ptr(ws),
             ws.size, None)
    assert np.isfinite(g).all()
    views, off = [], 0
    for gr in grads:
        views.append(g[off:off + gr.numel()].reshape(gr.shape))
        off += gr.numel()
    assert off == g.size
    check_grads(views, sd, x, labels, keep, grads)

    # eval mode: running statistics, no dropout -- well conditioned, compared directly
    elog = np.zeros((B, C), np.float32)
    lib.call("howl_mobilenet_fwd", ptr(flat), ptr(bufs), C, ptr(xn), sb, sm, st, B, M, T, 0, None, 1.0, ptr(elog), ptr(ws),
             ws.size, None)
    esd = {k: v.clone() for k, v in sd.items()}
    off = 0
    for l in om.layer_table():
        for name in (".running_mean", ".running_var"):
            n = esd[l["bn"] + name].numel()
            esd[l["bn"] + name] = torch.from_numpy(bufs[off:off + n].copy())
            off += n
    eref = om.mobilenet_forward(esd, x, False)
    np.testing.assert_allclose(elog, eref.numpy(), rtol=0, atol=5e-5)
