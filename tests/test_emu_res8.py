"""res8 forward / backward / loss / AdamW kernels on the hipemu CPU emulator vs the oracle (tiny batches).
Same C-ABI calls the GPU path makes; catches layout, fragment-mapping, halo and reduction mistakes on CPU."""
import ctypes

import numpy as np
import pytest
import torch

from emu_util import emu_lib, ptr
from howl_amd.lib import HowlAdamW, HowlRes8Grads, HowlRes8Params, HowlRes8Saved
from oracle import models as om


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


class Res8Harness:
    """Owns numpy buffers for params / saved activations / grads and fills the C structs."""

    def __init__(self, lib, B, T, C, sd=None, M=40):
        self.lib, self.B, self.T, self.C = lib, B, T, C
        self.H = T // 3
        W = M // 4      # pooled columns (80 mel bins: the library's own two-strip layout inside the same number of floats)
        self.sd = sd or om.res8_init(C)
        self.np = {k: np.ascontiguousarray(v.numpy()) for k, v in self.sd.items()}
        self.prm = HowlRes8Params()
        self.prm.conv0_w = ptr(self.np["conv0.weight"])
        for i in range(6):
            self.prm.conv_w[i] = ptr(self.np[f"conv{i+1}.weight"]).value
            self.prm.bn_running_mean[i] = ptr(self.np[f"bn{i+1}.running_mean"]).value
            self.prm.bn_running_var[i] = ptr(self.np[f"bn{i+1}.running_var"]).value
            self.prm.bn_num_batches[i] = ptr(self.np[f"bn{i+1}.num_batches_tracked"]).value
        self.prm.out_w = ptr(self.np["output.weight"])
        self.prm.out_b = ptr(self.np["output.bias"])
        nsaved = lib.cdll.howl_res8_saved_floats(B, T, M)
        assert nsaved >= B * 45 * self.H * W and (T > 83 or nsaved == B * 45 * self.H * W)
        self.s = [np.full(nsaved, np.nan, np.float32) for _ in range(7)]
        self.bn_stats = np.zeros((6, 2, 48), np.float32)
        self.pooled = np.zeros((B, 48), np.float32)
        self.saved = HowlRes8Saved()
        for i in range(7):
            self.saved.s[i] = ptr(self.s[i]).value
        self.saved.bn_stats = ptr(self.bn_stats)
        self.saved.pooled = ptr(self.pooled)
        self.mask0 = np.zeros(nsaved, np.uint16)
        self.saved.mask0 = ptr(self.mask0)
        self.grads_np = {k: np.full_like(self.np[k], np.nan) for k in om.res8_param_names()}
        self.gr = HowlRes8Grads()
        self.gr.conv0_w = ptr(self.grads_np["conv0.weight"])
        for i in range(6):
            self.gr.conv_w[i] = ptr(self.grads_np[f"conv{i+1}.weight"]).value
        self.gr.out_w = ptr(self.grads_np["output.weight"])
        self.gr.out_b = ptr(self.grads_np["output.bias"])
        nbytes = lib.cdll.howl_res8_workspace_bytes_mels(B, T, M)
        assert M != 40 or nbytes == lib.cdll.howl_res8_workspace_bytes(B, T)
        self.ws = np.zeros(nbytes, np.uint8)
        self.logits = np.zeros((B, C), np.float32)

    def fwd(self, feat_btm, training):
        f = np.ascontiguousarray(feat_btm, np.float32)   # (B, T, M)
        self._feat = f
        B, T, M = f.shape
        self.lib.call("howl_res8_fwd", ctypes.byref(self.prm), ptr(f), T * M, M, 1, B, T, M, self.C, int(training),
                      ctypes.byref(self.saved), ptr(self.logits), ptr(self.ws), self.ws.size, None)
        return self.logits.copy()

    def bwd(self, dlogits):
        d = np.ascontiguousarray(dlogits, np.float32)
        f = self._feat
        B, T, M = f.shape
        self.lib.call("howl_res8_bwd", ctypes.byref(self.prm), ptr(f), T * M, M, 1, B, T, M, self.C,
                      ctypes.byref(self.saved), ptr(d), ctypes.byref(self.gr), ptr(self.ws), self.ws.size, None)
        return {k: v.copy() for k, v in self.grads_np.items()}


def feats(B, T, seed, M=40):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, 3, M, T)).astype(np.float32)
    return torch.from_numpy(x)


@pytest.mark.parametrize("B,T,C", [(2, 81, 12), (3, 41, 4), (2, 20, 4)])
def test_res8_eval_forward(lib, B, T, C):
    x = feats(B, T, 0)
    sd = om.res8_init(C)
    for i in range(1, 7):   # non-trivial running statistics
        sd[f"bn{i}.running_mean"] = 0.1 * torch.arange(45, dtype=torch.float32).sin()
        sd[f"bn{i}.running_var"] = 0.5 + 0.3 * torch.arange(45, dtype=torch.float32).cos() ** 2
    h = Res8Harness(lib, B, T, C, sd={k: v.clone() for k, v in sd.items()})
    out = h.fwd(x[:, 0].permute(0, 2, 1).numpy(), training=False)
    ref = om.res8_forward(sd, x, False).numpy()
    np.testing.assert_allclose(out, ref, rtol=0, atol=2e-5)
    assert np.array_equal(out.argmax(1), ref.argmax(1))


@pytest.mark.parametrize("B,T,C", [(2, 81, 12), (3, 41, 4), (2, 20, 4)])
def test_res8_train_step(lib, B, T, C):
    x = feats(B, T, 1)
    labels = torch.arange(B) % C
    h = Res8Harness(lib, B, T, C)
    sd = om.res8_init(C)
    logits = h.fwd(x[:, 0].permute(0, 2, 1).numpy(), training=True)

    # oracle forward/backward with autograd, capturing the same intermediates
    names = om.res8_param_names()
    params = {n: sd[n].clone().requires_grad_(True) for n in names}
    sd_ref = dict(sd)
    sd_ref.update(params)
    ref_logits = om.res8_forward(sd_ref, x, True)
    np.testing.assert_allclose(logits, ref_logits.detach().numpy(), rtol=0, atol=2e-5)
    for i in (1, 3, 6):
        np.testing.assert_allclose(h.np[f"bn{i}.running_mean"], sd_ref[f"bn{i}.running_mean"].numpy(), atol=1e-6)
        np.testing.assert_allclose(h.np[f"bn{i}.running_var"], sd_ref[f"bn{i}.running_var"].numpy(), atol=1e-6)
        assert h.np[f"bn{i}.num_batches_tracked"] == 1

    # fused cross-entropy
    loss = np.zeros(1, np.float32)
    dlogits = np.zeros((B, C), np.float32)
    lab = np.ascontiguousarray(labels.numpy(), np.int64)
    lib.call("howl_xent_fwd_bwd", ptr(h.logits), ptr(lab), B, C, ptr(loss), ptr(dlogits), None)
    ref_loss = torch.nn.functional.cross_entropy(ref_logits, labels)
    grads_ref = torch.autograd.grad(ref_loss, [params[n] for n in names])
    assert abs(loss[0] - ref_loss.item()) < 1e-5

    grads = h.bwd(dlogits)
    for n, gref in zip(names, grads_ref):
        scale = max(1.0, float(gref.abs().max()))
        np.testing.assert_allclose(grads[n], gref.numpy(), rtol=0, atol=2e-5 * scale, err_msg=n)


def _wide_step(lib, B, T, C, seed, M=80):
    """One training step at M mel bins on the emulator and on the oracle: (harness, logits, grads, oracle logits, oracle grads)."""
    x = feats(B, T, seed, M=M)
    labels = torch.arange(B) % C
    h = Res8Harness(lib, B, T, C, M=M)
    sd = om.res8_init(C)
    logits = h.fwd(x[:, 0].permute(0, 2, 1).numpy(), training=True)
    names = om.res8_param_names()
    params = {n: sd[n].clone().requires_grad_(True) for n in names}
    sd_ref = dict(sd)
    sd_ref.update(params)
    ref_logits = om.res8_forward(sd_ref, x, True)
    loss, dlogits = np.zeros(1, np.float32), np.zeros((B, C), np.float32)
    lab = np.ascontiguousarray(labels.numpy(), np.int64)
    lib.call("howl_xent_fwd_bwd", ptr(h.logits), ptr(lab), B, C, ptr(loss), ptr(dlogits), None)
    grads = h.bwd(dlogits)
    gref = torch.autograd.grad(torch.nn.functional.cross_entropy(ref_logits, labels), [params[n] for n in names])
    return h, sd_ref, logits, grads, ref_logits.detach().numpy(), dict(zip(names, gref))


@pytest.mark.parametrize("B,T,C", [(2, 81, 12), (1, 20, 4)])
def test_res8_at_80_mel_bins_train_step(lib, B, T, C):
    """NUM_MELS = 80 (the reference's stock default, settings.py:32; cnn.py:113-145 pools (3,4) over any width): 20 pooled columns
    run as two strips of 10 that fetch each other's edge column in all three 3x3 roles (csrc/res8.hip HaloSlot).  Logits,
    BatchNorm buffers and every gradient against the oracle; odd batches and a grid of one strip pair included."""
    h, sd_ref, logits, grads, ref_logits, gref = _wide_step(lib, B, T, C, seed=21)
    np.testing.assert_allclose(logits, ref_logits, rtol=0, atol=2e-5)
    for i in (1, 3, 6):
        np.testing.assert_allclose(h.np[f"bn{i}.running_mean"], sd_ref[f"bn{i}.running_mean"].numpy(), atol=1e-6)
        np.testing.assert_allclose(h.np[f"bn{i}.running_var"], sd_ref[f"bn{i}.running_var"].numpy(), atol=1e-6)
    for n, g in gref.items():
        scale = max(1.0, float(g.abs().max()))
        np.testing.assert_allclose(grads[n], g.numpy(), rtol=0, atol=2e-5 * scale, err_msg=n)


@pytest.mark.parametrize("B,T,C,M", [(2, 101, 4, 40), (1, 101, 4, 80), (1, 250, 12, 40)])
def test_res8_trains_beyond_83_frames(lib, B, T, C, M):
    """cnn.py:127-145 takes any T.  More than 27 pooled rows do not fit the kernels' tile, so a longer utterance runs as row strips
    of equal height that fetch real halo rows (and corners) from their neighbours; the last strip may own fewer rows than its
    block has -- those are zero on the way into every tile and left out of every sum (csrc/res8.hip StripGeom).  101 frames: two
    strips of 17 rows, the second with 16; the same at 80 bins: 2 x 2 strips with corners; 250: four strips of 21 (20).  Logits, BatchNorm buffers
    and every gradient against the oracle."""
    h, sd_ref, logits, grads, ref_logits, gref = _wide_step(lib, B, T, C, seed=31, M=M)
    np.testing.assert_allclose(logits, ref_logits, rtol=0, atol=2e-5)
    for i in (1, 3, 6):
        np.testing.assert_allclose(h.np[f"bn{i}.running_mean"], sd_ref[f"bn{i}.running_mean"].numpy(), atol=1e-6)
        np.testing.assert_allclose(h.np[f"bn{i}.running_var"], sd_ref[f"bn{i}.running_var"].numpy(), atol=1e-6)
    for n, g in gref.items():
        scale = max(1.0, float(g.abs().max()))
        np.testing.assert_allclose(grads[n], g.numpy(), rtol=0, atol=2e-5 * scale, err_msg=n)


def test_golden_two_second_windows_on_the_emulator(lib, golden):
    """G14 (outputs of the reference's res8 on 161-frame inputs: two row strips of 27 + 26 pooled rows per utterance): the kernels
    on the emulator against the reference's training logits, loss gradient of every parameter and BatchNorm buffers, 40 mel bins."""
    g = golden("g14_res8_two_second_windows")
    x = g["m40.x"]                                   # (3, 1, 40, 161)
    B, C, T = 3, 12, 161
    h = Res8Harness(lib, B, T, C)
    logits = h.fwd(np.ascontiguousarray(x[:, 0].transpose(0, 2, 1)), training=True)
    np.testing.assert_allclose(logits, g["m40.train_logits"], rtol=0, atol=2e-5)
    loss, dlogits = np.zeros(1, np.float32), np.zeros((B, C), np.float32)
    lab = (np.arange(B) % C).astype(np.int64)
    lib.call("howl_xent_fwd_bwd", ptr(h.logits), ptr(lab), B, C, ptr(loss), ptr(dlogits), None)
    assert abs(loss[0] - float(g["m40.loss0"])) < 1e-5
    grads = h.bwd(dlogits)
    for n in om.res8_param_names():
        ref = g["m40.grad0." + n]
        np.testing.assert_allclose(grads[n], ref, rtol=0, atol=2e-5 * max(1.0, float(np.abs(ref).max())), err_msg=n)
    for i in (1, 6):
        np.testing.assert_allclose(h.np[f"bn{i}.running_mean"], g[f"m40.bn{i}.running_mean.1"], atol=1e-6)
        np.testing.assert_allclose(h.np[f"bn{i}.running_var"], g[f"m40.bn{i}.running_var.1"], atol=1e-6)


@pytest.mark.parametrize("B,T", [(2, 81), (1, 41)])
def test_res8_at_80_mel_bins_eval_forward(lib, B, T):
    C = 12
    x = feats(B, T, 4, M=80)
    sd = om.res8_init(C)
    for i in range(1, 7):
        sd[f"bn{i}.running_mean"] = 0.1 * torch.arange(45, dtype=torch.float32).sin()
        sd[f"bn{i}.running_var"] = 0.5 + 0.3 * torch.arange(45, dtype=torch.float32).cos() ** 2
    h = Res8Harness(lib, B, T, C, sd={k: v.clone() for k, v in sd.items()}, M=80)
    out = h.fwd(x[:, 0].permute(0, 2, 1).numpy(), training=False)
    ref = om.res8_forward(sd, x, False).numpy()
    np.testing.assert_allclose(out, ref, rtol=0, atol=2e-5)
    # the strided (B, M, T) view of the frontend's output gives the same bits
    g = np.ascontiguousarray(x[:, 0].numpy())
    out2 = np.full_like(out, np.nan)
    lib.call("howl_res8_fwd", ctypes.byref(h.prm), ptr(g), 80 * T, 1, T, B, T, 80, C, 0, ctypes.byref(h.saved), ptr(out2), ptr(h.ws),
             h.ws.size, None)
    np.testing.assert_array_equal(out2, out)


def test_res8_at_80_mel_bins_long_input(lib):
    """howl_res8_fwd_long at 80 mel bins: windows of 27 pooled rows x two strips each."""
    B, T, C = 1, 120, 12
    sd = om.res8_init(C)
    rng = np.random.default_rng(9)
    for i in range(1, 7):
        sd[f"bn{i}.running_mean"] = torch.from_numpy(rng.uniform(0.1, 0.6, 45).astype(np.float32))
        sd[f"bn{i}.running_var"] = torch.from_numpy(rng.uniform(0.5, 2.0, 45).astype(np.float32))
    h = Res8Harness(lib, 1, 81, C, sd=sd)
    x = feats(B, T, seed=10, M=80)
    ref = om.res8_forward({k: v.clone() for k, v in sd.items()}, x, False).numpy()
    f = np.ascontiguousarray(x[:, 0].permute(0, 2, 1).numpy())
    ws = np.zeros(lib.cdll.howl_res8_long_workspace_bytes_mels(B, T, 80), np.uint8)
    logits = np.full((B, C), np.nan, np.float32)
    lib.call("howl_res8_fwd_long", ctypes.byref(h.prm), ptr(f), T * 80, 80, 1, B, T, 80, C, ptr(logits), ptr(ws), ws.size, None)
    np.testing.assert_allclose(logits, ref, rtol=0, atol=1e-4 * max(1.0, float(np.abs(ref).max())))



def test_adamw(lib):
    rng = np.random.default_rng(3)
    n = 1000
    p = rng.standard_normal(n).astype(np.float32)
    m = np.zeros(n, np.float32)
    v = np.zeros(n, np.float32)
    tp = torch.from_numpy(p.copy()).requires_grad_(True)
    opt = torch.optim.AdamW([tp], 0.01, weight_decay=1e-2)
    for step in range(1, 4):
        g = rng.standard_normal(n).astype(np.float32)
        lib.call("howl_adamw_step", ptr(p), ptr(g), ptr(m), ptr(v), n, 0.01, 0.9, 0.999, 1e-8, 1e-2, step, 1.0, None)
        tp.grad = torch.from_numpy(g.copy())
        opt.step()
        np.testing.assert_allclose(p, tp.detach().numpy(), rtol=0, atol=2e-6)


@pytest.mark.parametrize("B,T", [(2, 84), (1, 120), (2, 201), (1, 164)])
def test_long_input_windows_vs_oracle(lib, B, T):
    """howl_res8_fwd_long: clips beyond the 83-frame on-chip map run as overlapping 27-row windows; the logits must equal the
    oracle's eval-mode forward over the whole clip (cnn.py:127-145 accepts any T).  T = 84: one row more than fits;
    T % 3 != 0: trailing frames that only conv0 sees; 201: five windows; 164: the last window overlaps its neighbour most."""
    C = 12
    sd = om.res8_init(C)
    rng = np.random.default_rng(T)
    for i in range(1, 7):      # non-trivial running statistics
        sd[f"bn{i}.running_mean"] = torch.from_numpy(rng.uniform(0.1, 0.6, 45).astype(np.float32))
        sd[f"bn{i}.running_var"] = torch.from_numpy(rng.uniform(0.5, 2.0, 45).astype(np.float32))
    h = Res8Harness(lib, 1, 81, C, sd=sd)           # parameter plumbing only
    x = feats(B, T, seed=T + 1)
    ref = om.res8_forward({k: v.clone() for k, v in sd.items()}, x, False).numpy()
    f = np.ascontiguousarray(x[:, 0].permute(0, 2, 1).numpy())      # (B, T, M)
    ws = np.zeros(lib.cdll.howl_res8_long_workspace_bytes(B, T), np.uint8)
    logits = np.full((B, C), np.nan, np.float32)
    lib.call("howl_res8_fwd_long", ctypes.byref(h.prm), ptr(f), T * 40, 40, 1, B, T, 40, C, ptr(logits), ptr(ws), ws.size, None)
    np.testing.assert_allclose(logits, ref, rtol=0, atol=1e-4 * max(1.0, float(np.abs(ref).max())))
    assert np.array_equal(logits.argmax(1), ref.argmax(1))
    # the same clip through the strided (B, 1, M, T) view the frontend hands out
    g = np.ascontiguousarray(x[:, 0].numpy())                        # (B, M, T)
    logits2 = np.full((B, C), np.nan, np.float32)
    lib.call("howl_res8_fwd_long", ctypes.byref(h.prm), ptr(g), 40 * T, 1, T, B, T, 40, C, ptr(logits2), ptr(ws), ws.size, None)
    assert np.array_equal(logits2, logits)


def test_small_batch_slicing_matches_unsliced():
    """Batches smaller than the CU count: several workgroups share an utterance (forward / data gradient: position tiles 2- or
    4-way, weight gradient: its N tiles 2-way; csrc/res8.hip conv_slices / pair_slices).  The same one-utterance step in a
    process whose emulated device has 8 CUs (forward 4-way, backward 4 + 2) and in one with a single CU (no slicing) must
    give the same logits, statistics and gradients up to summation order, and both must match the oracle."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    code = r"""
import json, os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
os.environ.setdefault("NUM_MELS", "40")
import numpy as np, torch
from emu_util import emu_lib, ptr
import test_emu_res8 as T
lib = emu_lib()
out = {}
for B, Tf, C in ((1, 81, 12), (2, 41, 4)):
    x = T.feats(B, Tf, 5)
    h = T.Res8Harness(lib, B, Tf, C)
    logits = h.fwd(x[:, 0].permute(0, 2, 1).numpy(), training=True)
    dlogits = np.zeros((B, C), np.float32); loss = np.zeros(1, np.float32)
    lab = np.ascontiguousarray((torch.arange(B) %% C).numpy(), np.int64)
    lib.call("howl_xent_fwd_bwd", ptr(h.logits), ptr(lab), B, C, ptr(loss), ptr(dlogits), None)
    g = h.bwd(dlogits)
    out["%%d_%%d" %% (B, Tf)] = {"logits": logits.tolist(), "grads": {k: v.reshape(-1)[:: max(1, v.size // 50)].tolist() for k, v in g.items()},
                               "rm": h.np["bn3.running_mean"].tolist()}
print("RESULT" + json.dumps(out))
""" % (str(Path(__file__).resolve().parent.parent), str(Path(__file__).resolve().parent))
    res = {}
    for cus in ("1", "8"):
        import os
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HIPEMU_CUS=cus), capture_output=True, text=True,
                           timeout=1500)
        assert r.returncode == 0, r.stderr[-2000:]
        res[cus] = json.loads(r.stdout.split("RESULT", 1)[1])
    for key in res["1"]:
        a, b = res["1"][key], res["8"][key]
        np.testing.assert_allclose(a["logits"], b["logits"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(a["rm"], b["rm"], rtol=0, atol=1e-6)
        for n in a["grads"]:
            ga, gb = np.asarray(a["grads"][n]), np.asarray(b["grads"][n])
            np.testing.assert_allclose(ga, gb, rtol=0, atol=2e-5 * max(1.0, np.abs(ga).max()), err_msg=n)
    # and the sliced run against the oracle
    x = feats(1, 81, 5)
    sd = om.res8_init(12)
    names = om.res8_param_names()
    params = {n: sd[n].clone().requires_grad_(True) for n in names}
    sd_ref = dict(sd)
    sd_ref.update(params)
    ref_logits = om.res8_forward(sd_ref, x, True)
    np.testing.assert_allclose(res["8"]["1_81"]["logits"], ref_logits.detach().numpy(), rtol=0, atol=2e-5)
    gref = torch.autograd.grad(torch.nn.functional.cross_entropy(ref_logits, torch.arange(1) % 12), [params[n] for n in names])
    for n, g in zip(names, gref):
        got = np.asarray(res["8"]["1_81"]["grads"][n])
        want = g.numpy().reshape(-1)[:: max(1, g.numel() // 50)]
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-5 * max(1.0, float(g.abs().max())), err_msg=n)


def test_fused_forward_cross_entropy_is_bit_identical(lib):
    """howl_res8_fwd_xent + howl_res8_bwd_xent (the loss inside the forward's last launch, the pooled gradient left in the
    workspace, the batch mean taken by the backward's head launch) == howl_res8_fwd + howl_xent_fwd_bwd + howl_res8_bwd, bit
    for bit: logits, loss, dlogits, every gradient and the BatchNorm buffers; both `part` conventions."""
    B, T, C = 3, 41, 12
    x = feats(B, T, 11)[:, 0].permute(0, 2, 1).numpy()
    labels = np.array([3, 0, 11], np.int64)
    ref = Res8Harness(lib, B, T, C)
    logits = ref.fwd(x, training=True)
    loss, dl = np.zeros(1, np.float32), np.zeros((B, C), np.float32)
    lib.call("howl_xent_fwd_bwd", ptr(ref.logits), ptr(labels), B, C, ptr(loss), ptr(dl), None)
    g_ref = ref.bwd(dl)
    for parts in ((0,), (1, 2)):
        h = Res8Harness(lib, B, T, C)
        f = np.ascontiguousarray(x, np.float32)
        nll, dl2, loss2 = np.full(B, np.nan, np.float32), np.full((B, C), np.nan, np.float32), np.full(1, np.nan, np.float32)
        lib.call("howl_res8_fwd_xent", ctypes.byref(h.prm), ptr(f), T * 40, 40, 1, B, T, 40, C, ctypes.byref(h.saved), ptr(labels),
                 ptr(h.logits), ptr(nll), ptr(dl2), ptr(h.ws), h.ws.size, None)
        for part in parts:
            lib.call("howl_res8_bwd_xent", ctypes.byref(h.prm), ptr(f), T * 40, 40, 1, B, T, 40, C, ctypes.byref(h.saved), ptr(dl2),
                     ptr(nll), ptr(loss2), ctypes.byref(h.gr), ptr(h.ws), h.ws.size, part, None, None)
        np.testing.assert_array_equal(h.logits, logits)
        np.testing.assert_array_equal(dl2, dl)
        np.testing.assert_array_equal(loss2, loss)
        for k, v in g_ref.items():
            np.testing.assert_array_equal(h.grads_np[k], v, err_msg=k)
        np.testing.assert_array_equal(h.np["bn3.running_mean"], ref.np["bn3.running_mean"])


@pytest.mark.parametrize("B", [3])
def test_optimiser_step_in_the_last_fold_launch(lib, B):
    """howl_res8_bwd_xent(..., HowlAdamW) == howl_res8_bwd_xent + howl_adamw_step on the same flat buffers, bit for bit (gradients,
    parameters, both moments): all seven weight-gradient rows fold in the last launch here, the step's extra row of blocks takes
    the head's parameters (the large-batch split -- layers 2..6 folded inside the pair launches, the extra row takes them too --
    needs >= 64 data-gradient workgroups: tests/test_gpu_res8.py::test_optimiser_step_in_the_fold_is_bit_identical_on_the_device).
    Gradient pointers that are NOT the flat buffer fall back to the optimiser's own launch behind the fold."""
    T, C = 30, 12
    x = feats(B, T, 3)[:, 0].permute(0, 2, 1).numpy()
    labels = (np.arange(B) % C).astype(np.int64)
    names = om.res8_param_names()
    rng = np.random.default_rng(2)

    def run(mode):     # "apart" | "fold" | "scattered" (fused call with gradient arrays outside the flat buffer)
        h = Res8Harness(lib, B, T, C)
        sizes = [h.np[k].size for k in names]
        offs = np.concatenate([[0], np.cumsum(sizes)])
        n = int(offs[-1])
        flat = np.concatenate([h.np[k].reshape(-1) for k in names]).astype(np.float32)
        g = np.full(n, np.nan, np.float32)
        m, v = np.linspace(-0.01, 0.01, n).astype(np.float32), np.linspace(1e-6, 1e-4, n).astype(np.float32)
        views = {k: g[offs[i]:offs[i + 1]] for i, k in enumerate(names)}
        if mode != "scattered":
            h.grads_np = views
            h.gr.conv0_w = ptr(views["conv0.weight"])
            for i in range(6):
                h.gr.conv_w[i] = ptr(views[f"conv{i+1}.weight"]).value
            h.gr.out_w, h.gr.out_b = ptr(views["output.weight"]), ptr(views["output.bias"])
        f = np.ascontiguousarray(x, np.float32)
        nll, dl, loss = np.full(B, np.nan, np.float32), np.full((B, C), np.nan, np.float32), np.full(1, np.nan, np.float32)
        lib.call("howl_res8_fwd_xent", ctypes.byref(h.prm), ptr(f), T * 40, 40, 1, B, T, 40, C, ctypes.byref(h.saved), ptr(labels),
                 ptr(h.logits), ptr(nll), ptr(dl), ptr(h.ws), h.ws.size, None)
        opt = HowlAdamW(ptr(flat), ptr(g), ptr(m), ptr(v), n, 0.01, 0.9, 0.999, 1e-8, 1e-5, 2, 1.0)
        if mode == "scattered":      # the step then reads g: give it the separately computed gradients
            lib.call("howl_res8_bwd_xent", ctypes.byref(h.prm), ptr(f), T * 40, 40, 1, B, T, 40, C, ctypes.byref(h.saved), ptr(dl), ptr(nll),
                     ptr(loss), ctypes.byref(h.gr), ptr(h.ws), h.ws.size, 0, None, None)
            for k in names:
                views[k][:] = h.grads_np[k].reshape(-1)
            lib.call("howl_res8_fwd_xent", ctypes.byref(h.prm), ptr(f), T * 40, 40, 1, B, T, 40, C, ctypes.byref(h.saved), ptr(labels),
                     ptr(h.logits), ptr(nll), ptr(dl), ptr(h.ws), h.ws.size, None)
        lib.call("howl_res8_bwd_xent", ctypes.byref(h.prm), ptr(f), T * 40, 40, 1, B, T, 40, C, ctypes.byref(h.saved), ptr(dl), ptr(nll),
                 ptr(loss), ctypes.byref(h.gr), ptr(h.ws), h.ws.size, 0, None if mode == "apart" else ctypes.byref(opt), None)
        if mode == "apart":
            lib.call("howl_adamw_step", ptr(flat), ptr(g), ptr(m), ptr(v), n, 0.01, 0.9, 0.999, 1e-8, 1e-5, 2, 1.0, None)
        return flat, g, m, v

    a, b_ = run("apart"), run("fold")
    for u, w in zip(a, b_):
        assert np.isfinite(u).all()
        np.testing.assert_array_equal(u, w)
    if B == 3:
        c = run("scattered")
        for u, w in zip((a[0], a[2], a[3]), (c[0], c[2], c[3])):
            np.testing.assert_array_equal(u, w)


def test_many_utterances_per_workgroup_fused_and_unfused_backward():
    """The tile staging runs UNDER the K loop (csrc/res8.hip, "Tile staging UNDER the K loop"): a workgroup with several
    utterances stages the next one's halves while it multiplies the current one's.  Two emulated CUs, so that workgroups carry
    2-5 utterances in the forward (conv grid 2) and in both roles of the backward pair (grid 1 each); geometries with an odd and
    an even number of weight-gradient rounds.  The fused backward (BatchNorm / ReLU backward built inside the pair's staging,
    default) and the unfused one (HOWL_RES8_BWD_FUSED=0: bn_relu_bwd_kernel writes dz) must agree with each other to rounding
    and both with the oracle."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    code = r"""
import json, os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
os.environ.setdefault("NUM_MELS", "40")
import numpy as np, torch
from emu_util import emu_lib, ptr
import test_emu_res8 as T
from oracle import models as om
lib = emu_lib()
out = {}
for B, Tf, C in ((5, 41, 4), (3, 62, 5)):
    x = T.feats(B, Tf, 7)
    labels = torch.arange(B) %% C
    sd = om.res8_init(C); names = om.res8_param_names()
    params = {n: sd[n].clone().requires_grad_(True) for n in names}
    sd_ref = dict(sd); sd_ref.update(params)
    ref_logits = om.res8_forward(sd_ref, x, True)
    gref = torch.autograd.grad(torch.nn.functional.cross_entropy(ref_logits, labels), [params[n] for n in names])
    res = {}
    for fused in ("1", "0"):
        os.environ["HOWL_RES8_BWD_FUSED"] = fused
        h = T.Res8Harness(lib, B, Tf, C)
        logits = h.fwd(x[:, 0].permute(0, 2, 1).numpy(), training=True)
        dlogits = np.zeros((B, C), np.float32); loss = np.zeros(1, np.float32)
        lab = np.ascontiguousarray(labels.numpy(), np.int64)
        lib.call("howl_xent_fwd_bwd", ptr(h.logits), ptr(lab), B, C, ptr(loss), ptr(dlogits), None)
        res[fused] = (logits, h.bwd(dlogits))
    errs = {"logits": float(np.abs(res["1"][0] - ref_logits.detach().numpy()).max())}
    for n, g in zip(names, gref):
        scale = max(1.0, float(g.abs().max()))
        errs["oracle." + n] = max(float(np.abs(res[f][1][n] - g.numpy()).max()) for f in ("1", "0")) / scale
        errs["fused_vs_unfused." + n] = float(np.abs(res["1"][1][n] - res["0"][1][n]).max()) / scale
    out["%%d_%%d" %% (B, Tf)] = errs
# 80 mel bins (two strips per utterance, HaloSlot): three utterances = six strips over a grid of two workgroups, so that every
# workgroup stages a NEXT strip's halo column under its K loop (forward, and both roles of the backward pair with grid 2)
os.environ.pop("HOWL_RES8_BWD_FUSED", None)
h, sd_ref, logits, grads, ref_logits, gref = T._wide_step(lib, 3, 41, 4, seed=8)
errs = {"logits": float(np.abs(logits - ref_logits).max())}
for n, g in gref.items():
    errs["oracle." + n] = float(np.abs(grads[n] - g.numpy()).max()) / max(1.0, float(g.abs().max()))
out["wide_3_41"] = errs
print("RESULT" + json.dumps(out))
""" % (str(Path(__file__).resolve().parent.parent), str(Path(__file__).resolve().parent))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HIPEMU_CUS="2"), capture_output=True, text=True, timeout=2400)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads(r.stdout.split("RESULT", 1)[1])
    for key, errs in res.items():
        for name, e in errs.items():
            tol = 2e-5 if name == "logits" or name.startswith("oracle.") else 2e-6
            assert e < tol, (key, name, e)
