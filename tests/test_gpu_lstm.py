"""-m gpu parity: lstm / seq-lstm on the HIP path vs golden vectors captured from the reference (G6) and the oracle at
the BASELINE config-4 size (seq-lstm, batch 512, 0.5 s windows, CTC)."""
import numpy as np
import pytest
import torch

from gpu_util import DEV, golden_features, maxerr, t
from oracle import frontend as ofe
from oracle import models as om

pytestmark = pytest.mark.gpu


def make(name, C):
    from howl_amd.model import RegisteredModel
    m = RegisteredModel.find_registered_class(name)(C)
    m.load_state_dict({k: v.clone() for k, v in om.lstm_init(C).items()})
    return m.to(DEV)


@pytest.mark.parametrize("rows", ["4", "16"])
@pytest.mark.parametrize("name", ["lstm", "seq-lstm"])
def test_golden_forward_backward_streaming(golden, name, rows, monkeypatch):
    monkeypatch.setenv("HOWL_LSTM_ROWS", rows)        # both recurrence pairs (4 / 16 sequences per workgroup)
    g = golden("g6_" + name.replace("-", "_"))
    x, _ = golden_features(golden)
    xd = x.to(DEV)
    flen = t(g["frame_lengths"])
    model = make(name, 5).eval()
    with torch.no_grad():
        logits = model(xd, flen)
    assert logits.shape == g["logits"].shape
    assert maxerr(logits, g["logits"]) < 1e-3 and maxerr(logits, g["logits"]) < 2e-5
    model.train()
    sc = model(xd, flen)
    if name == "lstm":
        loss = torch.nn.functional.cross_entropy(sc, (torch.arange(6) % 5).to(DEV))
    else:
        lp = torch.nn.functional.log_softmax(sc, -1)
        loss = torch.nn.CTCLoss(4)(lp, torch.tensor([[0, 1, 2]] * 6).to(DEV), flen, torch.tensor([3] * 6))
    loss.backward()
    assert abs(loss.item() - float(g["loss0"])) < 1e-4
    for n, p in model.named_parameters():
        ref = g["grad0." + n]
        assert maxerr(p.grad, ref) < 5e-5 * max(1.0, float(np.abs(ref).max())), n
    # streaming carry over two consecutive calls (rnn.py:62,67-68)
    model.eval().streaming()
    with torch.no_grad():
        if name == "seq-lstm":
            a = model(xd[:1, :, :, :40], None)
            b = model(xd[:1, :, :, 40:], None)
        else:
            a = model(xd[:1, :, :, :40], torch.tensor([40]))
            b = model(xd[:1, :, :, 40:], torch.tensor([41]))
    assert maxerr(a, g["stream_a"]) < 2e-5 and maxerr(b, g["stream_b"]) < 2e-5
    with pytest.raises(RuntimeError):
        model(xd, torch.tensor([10, 78, 78, 78, 69, 62]))        # unsorted lengths: same error class as pack_padded_sequence


def test_config4_seq_lstm_ctc_step_vs_oracle():
    """BASELINE configs[3]: seq-lstm, batch 512, 0.5 s (T=41, 38 valid frames), CTC blank=4; ragged variant included."""
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    from howl_amd.utils.synth import synthetic_pcm
    B, L, C = 512, 8000, 5
    pcm = synthetic_pcm(B, L)
    std = StandardAudioTransform().to(DEV).eval()
    zmuv = ZmuvTransform().to(DEV)
    zmuv.update(std(pcm[:4].to(DEV)))
    feats = std.log_mel_for_model(pcm.to(DEV), zmuv)
    fb = ofe.mel_fb(40)
    z = ofe.Zmuv()
    z.update(ofe.standard_audio_transform(pcm[:4], fb))
    x_ref = z(ofe.standard_audio_transform(pcm, fb))
    targets = torch.tensor([[0, 1, 2]] * B)
    import os
    for lengths, rows in ((torch.full((B,), 38), "4"), (torch.sort(20 + torch.arange(B) % 19, descending=True).values, "4"),
                          (torch.sort(20 + torch.arange(B) % 19, descending=True).values, "16")):
        os.environ["HOWL_LSTM_ROWS"] = rows
        model = make("seq-lstm", C).train()
        sc = model(feats, lengths)
        from howl_amd import ops
        loss = ops.ctc_loss(sc, targets, lengths, torch.tensor([3] * B), 4)     # fused log_softmax + CTCLoss(blank=4)
        loss.backward()
        sd = {k: v.clone().requires_grad_(True) for k, v in om.lstm_init(C).items()}
        ref, _ = om.seq_lstm_forward(sd, x_ref, lengths)
        ref_loss = torch.nn.CTCLoss(4)(torch.log_softmax(ref, -1), targets, lengths, torch.tensor([3] * B))
        ref_loss.backward()
        assert sc.shape == ref.shape and maxerr(sc, ref) < 1e-3
        assert torch.equal(sc.argmax(-1).cpu(), ref.argmax(-1))
        assert abs(loss.item() - ref_loss.item()) < 1e-4
        for n, p in model.named_parameters():
            r = sd[n].grad
            assert maxerr(p.grad, r) < 1e-4 * max(1.0, r.abs().max().item()), n
    os.environ.pop("HOWL_LSTM_ROWS", None)


@pytest.mark.parametrize("T,B,C,Lmax,seed", [(38, 64, 5, 3, 0), (81, 33, 12, 8, 1), (128, 16, 64, 31, 2), (9, 5, 3, 2, 3)])
def test_fused_ctc_loss_vs_torch_cpu(T, B, C, Lmax, seed):
    """ops.ctc_loss (one fused log_softmax + CTC forward/backward kernel) against the reference's arithmetic, torch's CPU
    log_softmax + ctc_loss + autograd: ragged input lengths, repeated labels, empty targets, the kernel's maximum sizes,
    host-side length vectors as in train.py; the (T,B,C) scores are the permuted view of a (B,T,C) buffer like the model's."""
    from ctc_util import make_case, reference
    from howl_amd import ops
    logits, targets, in_len, tgt_len, blank = make_case(T, B, C, Lmax, seed, tight=True)
    per, loss, grad = reference(logits, targets, in_len, tgt_len, blank)
    z = logits.to(DEV).requires_grad_(True)
    out = ops.ctc_loss(z.permute(1, 0, 2), targets, in_len, tgt_len, blank)
    out.backward()
    assert abs(out.item() - loss.item()) < 2e-5 * max(1.0, abs(loss.item()))
    assert maxerr(z.grad, grad) < 1e-5
    for b in range(B):
        assert not z.grad[b, int(in_len[b]):].any()
    # upstream gradient scaling and bit-identical repeats
    z2 = logits.to(DEV).requires_grad_(True)
    (3.0 * ops.ctc_loss(z2.permute(1, 0, 2), targets.to(DEV), in_len.to(DEV), tgt_len.to(DEV), blank)).backward()
    assert torch.equal(z2.grad, 3.0 * z.grad)


@pytest.mark.parametrize("T,B,C,Lmax,seed", [(300, 16, 5, 3, 4), (129, 7, 12, 8, 5), (512, 8, 5, 3, 6), (640, 4, 64, 31, 7),
                                             (1000, 3, 6, 5, 8)])
def test_fused_ctc_whole_clips_vs_torch_cpu(T, B, C, Lmax, seed):
    """Round 6: beyond 128 frames the same kernel walks the utterance in 128-frame windows (alpha rows of the leading windows
    parked in a workspace, beta back through them), so whole clips -- what AudioSequenceBatchifier batches
    (batchifier.py:14-34) and CTCLoss sees (train.py:291-296) -- stay on the library's kernel.  alpha + beta reach -700 .. -2000
    at these lengths (an fp32 ulp there is 6e-5 .. 1.2e-4): torch's own fp32 kernel sits that far from its fp64 run, and so does
    this one -- the bound is torch-fp32's own distance from fp64."""
    from ctc_util import make_case, reference
    from howl_amd import ops
    logits, targets, in_len, tgt_len, blank = make_case(T, B, C, Lmax, seed)
    in_len[-1] = 128                          # exactly one window, next to utterances of two to eight
    if B > 3 and T > 256:
        in_len[-2] = 256
    per, loss, grad = reference(logits, targets, in_len, tgt_len, blank)
    _, loss64, grad64 = reference(logits.double(), targets, in_len, tgt_len, blank)
    noise = float((grad.double() - grad64).abs().max())
    z = logits.to(DEV).requires_grad_(True)
    out = ops.ctc_loss(z.permute(1, 0, 2), targets, in_len, tgt_len, blank)
    out.backward()
    assert abs(out.item() - loss64.item()) < 2e-5 * max(1.0, abs(loss64.item()))
    assert maxerr(z.grad, grad64) < 1.5 * noise + 1e-5, (maxerr(z.grad, grad64), noise)
    for b in range(B):
        assert not z.grad[b, int(in_len[b]):].any()
    z2 = logits.to(DEV).requires_grad_(True)       # bit-identical repeats, device-resident lengths
    ops.ctc_loss(z2.permute(1, 0, 2), targets.to(DEV), in_len.to(DEV), tgt_len.to(DEV), blank).backward()
    assert torch.equal(z2.grad, z.grad)
    # an utterance of <= 128 frames: the same bits in a one-window launch
    short = int((in_len <= 128).nonzero()[0])
    z3 = logits[short:short + 1, :128].contiguous().to(DEV).requires_grad_(True)
    o3 = ops.ctc_loss(z3.permute(1, 0, 2), targets[short:short + 1], in_len[short:short + 1], tgt_len[short:short + 1], blank)
    o3.backward()
    assert torch.equal(z3.grad[0, :128] * (1.0 / B), z.grad[short, :128]) or \
        maxerr(z3.grad[0, :128] / B, z.grad[short, :128]) < 1e-9        # (the 1 / B of the batch mean is applied inside)


def test_fused_ctc_out_of_range_raises():
    """No vendor kernels on the training path: outside howl_ctc_loss's range (C > 64, a target of more than 31 labels, more than
    8192 frames) ops.ctc_loss raises instead of calling torch's ctc_loss; CPU scores raise as before (no CPU fallback)."""
    from ctc_util import make_case
    from howl_amd import ops
    from howl_amd.lib import HowlHipError
    logits, targets, in_len, tgt_len, blank = make_case(20, 4, 70, 3, 4)
    with pytest.raises(HowlHipError, match="range"):
        ops.ctc_loss(logits.to(DEV).permute(1, 0, 2), targets, in_len, tgt_len, blank)
    logits, targets, in_len, tgt_len, blank = make_case(40, 2, 5, 3, 4)
    with pytest.raises(HowlHipError, match="range"):
        ops.ctc_loss(logits.to(DEV).permute(1, 0, 2), torch.zeros(2, 40, dtype=torch.int64), in_len, torch.tensor([40, 3]), blank)
    with pytest.raises(Exception):
        ops.ctc_loss(logits.permute(1, 0, 2), targets, in_len, tgt_len, blank)   # CPU scores: no CPU fallback


@pytest.mark.parametrize("rows", ["4", "16"])
@pytest.mark.parametrize("name", ["lstm", "seq-lstm"])
def test_golden_whole_clips(golden, name, rows, monkeypatch):
    """G15: the reference's recurrent models on whole clips of 318 / 258 / 206 / 128 frames (eight times G6's 41; three, three,
    two and exactly one window of the CTC kernel): eval logits, one training step's loss and gradients through the library's
    frontend, recurrences, head and CTC kernel, and the streaming carry over a 160 + 161-frame split (rnn.py:60-71)."""
    monkeypatch.setenv("HOWL_LSTM_ROWS", rows)
    from howl_amd import ops
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    g = golden("g15_whole_clips_" + name.replace("-", "_"))
    x = t(g["x"]).to(DEV)
    flen = t(g["frame_lengths"])
    model = make(name, 5).eval()
    with torch.no_grad():
        logits = model(x, flen)
    assert logits.shape == g["logits"].shape and maxerr(logits, g["logits"]) < 5e-5
    model.train()
    sc = model(x, flen)
    if name == "lstm":
        loss = torch.nn.functional.cross_entropy(sc, (torch.arange(4) % 5).to(DEV))
    else:
        loss = ops.ctc_loss(sc, t(g["targets"]), flen, t(g["target_lengths"]), 4)
    loss.backward()
    assert abs(loss.item() - float(g["loss0"])) < 1e-4 * max(1.0, float(g["loss0"]))
    for n, p in model.named_parameters():
        ref = g["grad0." + n]
        assert maxerr(p.grad, ref) < 1e-4 * max(1.0, float(np.abs(ref).max())), n      # BPTT over 318 steps in fp32
    model.eval().streaming()
    with torch.no_grad():
        if name == "seq-lstm":
            a, b = model(x[:1, :, :, :160], None), model(x[:1, :, :, 160:], None)
        else:
            a, b = model(x[:1, :, :, :160], torch.tensor([160])), model(x[:1, :, :, 160:], torch.tensor([161]))
    assert maxerr(a, g["stream_a"]) < 5e-5 and maxerr(b, g["stream_b"]) < 5e-5
    if name == "seq-lstm":       # the library's frontend on the same clips -> the reference's features
        gz = golden("g4_zmuv")
        std = StandardAudioTransform().to(DEV).eval()
        zmuv = ZmuvTransform().to(DEV)
        zmuv.load_state_dict({"total": t(gz["total"]), "mean": t(gz["mean"]), "mean2": t(gz["mean2"])}, strict=False)
        feats = std.log_mel_for_model(t(g["audio"]).to(DEV), zmuv)
        assert feats.shape == x.shape and maxerr(feats, x) < 2e-3
        with torch.no_grad():        # (a fresh module: the one above is in streaming mode and carries a one-clip state)
            assert maxerr(make(name, 5).eval()(feats, flen), g["logits"]) < 1e-3


@pytest.mark.parametrize("B,rows", [(64, "4"), (64, "16"), (16, "4")])
def test_seq_lstm_whole_clip_ctc_step_vs_oracle(B, rows, monkeypatch):
    """The sequence objective as the reference feeds it: a ragged batch of WHOLE CLIPS (frame lengths 40 .. 318 sorted descending,
    targets of 1 .. 3 labels) through FusedTrainer.step_sequence -- frontend, forward recurrence, head, the windowed CTC kernel,
    one-call backward, AdamW in the fold -- against the oracle's step on the same audio (batchifier.py:14-34,
    train.py:198-200,291-296).  B = 16 is the reference's own batch size for this objective (envs/seq-lstm.env)."""
    monkeypatch.setenv("HOWL_LSTM_ROWS", rows)
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    from howl_amd.training.fused import FusedTrainer
    from howl_amd.utils.synth import synthetic_pcm
    L, C = 64000, 5
    pcm = synthetic_pcm(B, L)
    samples = torch.sort(torch.linspace(8400, L, B).long(), descending=True).values
    for b in range(B):
        pcm[b, samples[b]:] = 0.0                   # operator.py:77-86: zero-padded to the longest
    std = StandardAudioTransform().to(DEV).eval()
    zmuv = ZmuvTransform().to(DEV)
    zmuv.update(std(pcm[:4].to(DEV)))
    flen = std.compute_lengths(samples)
    assert int(flen.max()) == 318 and int(flen.min()) == 40
    targets = torch.tensor([[0, 1, 2], [3, 3, 0], [2, 1, 0], [1, 0, 3]] * (B // 4))
    tl = torch.tensor([3, 2, 1, 3] * (B // 4))
    model = make("seq-lstm", C).train()
    tr = FusedTrainer(model, std, zmuv, lr=1e-3, weight_decay=1e-5)
    loss = tr.step_sequence(pcm.to(DEV), flen, targets, tl, 4)
    grads = [gg.clone() for gg in tr.fp.grad_views]
    fb = ofe.mel_fb(40)
    z = ofe.Zmuv()
    z.update(ofe.standard_audio_transform(pcm[:4], fb))
    x_ref = z(ofe.standard_audio_transform(pcm, fb))
    # BPTT over up to 318 steps x B sequences behind a log-space CTC recursion whose alpha + beta reach -700: fp32 rounding alone
    # moves these gradients at the 1e-4 level (the fp32 oracle sits 7e-6 (W_ih) .. 8e-5 (dnn.0.bias) .. 1e-3 (dnn.2.bias, |g| <= 24)
    # from its own fp64 run on the B = 64 batch).  So the yardstick is the oracle itself: the fp64 oracle is the truth, the fp32
    # oracle's distance from it the noise scale, and the kernels must sit within a small multiple of that scale.
    refs = {}
    for dt in (torch.float32, torch.float64):
        sd = {k: v.clone().to(dt).requires_grad_(True) for k, v in om.lstm_init(C).items()}
        ref, _ = om.seq_lstm_forward(sd, x_ref.to(dt), flen)
        ref_loss = torch.nn.CTCLoss(4)(torch.log_softmax(ref, -1), targets, flen, tl)
        ref_loss.backward()
        refs[dt] = (ref.detach(), ref_loss.detach(), {k: v.grad for k, v in sd.items()})
    ref, ref_loss, g64 = refs[torch.float64]
    assert tr.last_logits.shape == ref.shape and maxerr(tr.last_logits, ref) < 1e-3
    assert abs(loss.item() - ref_loss.item()) < 1e-4 * max(1.0, ref_loss.item())
    for n, gg in zip(om.lstm_param_names(), grads):
        noise = maxerr(refs[torch.float32][2][n], g64[n])
        assert maxerr(gg, g64[n]) < 4.0 * noise + 2e-5 * max(1.0, g64[n].abs().max().item()), (n, maxerr(gg, g64[n]), noise)
    # bit-repeatable, and a second step moves the loss
    model2 = make("seq-lstm", C).train()
    tr2 = FusedTrainer(model2, std, zmuv, lr=1e-3, weight_decay=1e-5)
    loss_b = tr2.step_sequence(pcm.to(DEV), flen, targets, tl, 4)
    assert torch.equal(loss_b, loss) and all(torch.equal(a, b) for a, b in zip(grads, tr2.fp.grad_views))
    assert tr.step_sequence(pcm.to(DEV), flen, targets, tl, 4).item() < loss.item()


@pytest.mark.parametrize("B", [48, 512])
def test_fused_sequence_step_matches_autograd_path_and_oracle(B):
    """FusedTrainer.step_sequence (explicit launches, gradients straight into the flat buffer, flat AdamW) against (a) the
    autograd path through the same kernels + torch.optim.AdamW and (b) the oracle's CTC step; ragged lengths.  B = 512 is the
    size at which the library switches to its row-streaming head kernels and the trainer's one-call backward
    (howl_seq_lstm_bwd: job-array weight gradients) meets the autograd path's two calls: still bit-identical."""
    from howl_amd import ops
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    from howl_amd.training.fused import FusedTrainer
    from howl_amd.utils.synth import synthetic_pcm
    L, C = 8000, 5
    pcm = synthetic_pcm(B, L).to(DEV)
    std = StandardAudioTransform().to(DEV).eval()
    zmuv = ZmuvTransform().to(DEV)
    zmuv.update(std(pcm[:4]))
    lengths = torch.sort(24 + torch.arange(B) % 15, descending=True).values
    targets = torch.tensor([[0, 1, 2]] * B)
    tl = torch.tensor(([3, 2, 1] * (B // 3 + 1))[:B])
    feats = std.log_mel_for_model(pcm, zmuv)

    fused_model = make("seq-lstm", C).train()
    trainer = FusedTrainer(fused_model, std, zmuv, lr=1e-3, weight_decay=1e-5)
    loss_f = trainer.step_sequence(pcm, lengths, targets, tl, 4)
    grads_f = [g.clone() for g in trainer.fp.grad_views]

    ref_model = make("seq-lstm", C).train()
    opt = torch.optim.AdamW(ref_model.parameters(), 1e-3, weight_decay=1e-5)
    loss_a = ops.ctc_loss(ref_model(feats, lengths), targets, lengths, tl, 4)
    loss_a.backward()
    assert abs(loss_f.item() - loss_a.item()) < 1e-6
    # same kernels, same order: bit-identical -- except, from 2048 rows on (round 6), the three gradients whose per-workgroup
    # partial sums the trainer's fused head + CTC launch (howl_seq_head_ctc) takes over other rows than the autograd path's launches
    regrouped = ("dnn.0.bias", "dnn.2.weight", "dnn.2.bias") if fused_model.ctc_nll is not None else ()
    assert (fused_model.ctc_nll is not None) == (B * int(lengths.max()) >= 2048)
    for n, p, g in zip(om.lstm_param_names(), ref_model.hot_parameters(), grads_f):
        if n in regrouped:
            assert maxerr(p.grad, g) <= 2e-6 * max(1.0, g.abs().max().item()), n
        else:
            assert torch.equal(p.grad, g), n
    opt.step()
    for p, q in zip(ref_model.hot_parameters(), fused_model.hot_parameters()):
        assert maxerr(p, q) < 1e-6                                      # torch AdamW vs the flat kernel: rounding only

    sd = {k: v.clone().requires_grad_(True) for k, v in om.lstm_init(C).items()}
    fb = ofe.mel_fb(40)
    z = ofe.Zmuv()
    z.update(ofe.standard_audio_transform(pcm[:4].cpu(), fb))
    ref, _ = om.seq_lstm_forward(sd, z(ofe.standard_audio_transform(pcm.cpu(), fb)), lengths)
    ref_loss = torch.nn.CTCLoss(4)(torch.log_softmax(ref, -1), targets, lengths, tl)
    ref_loss.backward()
    assert abs(loss_f.item() - ref_loss.item()) < 1e-4
    names = ["lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0", "dnn.0.weight", "dnn.0.bias",
             "dnn.2.weight", "dnn.2.bias"]
    for n, g in zip(names, grads_f):
        r = sd[n].grad
        assert maxerr(g, r) < 1e-4 * max(1.0, r.abs().max().item()), n
    # a second step runs (saved state is per step) and the loss moves
    loss2 = trainer.step_sequence(pcm, lengths, targets, tl, 4)
    assert loss2.item() < loss_f.item()


def test_optimiser_step_in_the_fold_is_bit_identical_on_the_device(monkeypatch):
    """Round 5: howl_seq_lstm_bwd takes the AdamW step inside its slab fold on a single replica (HowlAdamW); three steps that way
    and three with the optimiser's own launch (HOWL_NO_FOLD_ADAMW=1) leave bit-identical parameters, moments and gradients --
    likewise with the head's weight gradient riding in the backward recurrence's launch (default) or behind it (HOWL_LSTM_RIDE=0)."""
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    from howl_amd.training.fused import FusedTrainer
    from howl_amd.utils.synth import synthetic_pcm
    B, L, C = 512, 8000, 5
    pcm = synthetic_pcm(B, L).to(DEV)
    std = StandardAudioTransform().to(DEV).eval()
    zmuv = ZmuvTransform().to(DEV)
    zmuv.update(std(pcm[:4]))
    lengths = torch.full((B,), 38)
    targets = torch.tensor([[0, 1, 2]] * B)
    tl = torch.tensor(([3, 2, 1] * (B // 3 + 1))[:B])
    out = []
    for env in ({}, {"HOWL_NO_FOLD_ADAMW": "1"}, {"HOWL_LSTM_RIDE": "0"}):
        for k in ("HOWL_NO_FOLD_ADAMW", "HOWL_LSTM_RIDE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        model = make("seq-lstm", C).train()
        tr = FusedTrainer(model, std, zmuv, lr=1e-3, weight_decay=1e-5)
        for _ in range(3):
            tr.step_sequence(pcm, lengths, targets, tl, 4)
        torch.cuda.synchronize()
        out.append([t.clone() for t in (tr.fp.flat, tr.m, tr.v, tr.fp.grad)])
    for other in out[1:]:
        for a, b in zip(out[0], other):
            assert torch.isfinite(a).all()
            assert torch.equal(a, b)


def test_look_ahead_frontend_rides_in_the_forward_call(monkeypatch):
    """Round 5: FusedTrainer.step_sequence(next_audio=...) -- the next batch's log-mel as rider blocks of this step's forward
    recurrence launch (howl_lstm_fwd_next).  Three steps over three DIFFERENT batches with the look-ahead, without it, and with the
    rider switched off inside the library (HOWL_LSTM_RIDE_LOGMEL=0: the frontend as its own launch behind the recurrence) leave
    bit-identical parameters, moments and losses; the features the rider wrote are the frontend's own, bit for bit; a batch that
    is NOT the one announced is recomputed (the look-ahead is dropped, never used for the wrong tensor)."""
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    from howl_amd.training.fused import FusedTrainer
    from howl_amd.utils.synth import synthetic_pcm
    B, L, C = 512, 8000, 5
    batches = [synthetic_pcm(B, L, seed=100 + i).to(DEV) for i in range(4)]
    std = StandardAudioTransform().to(DEV).eval()
    zmuv = ZmuvTransform().to(DEV)
    zmuv.update(std(batches[0][:4]))
    lengths = torch.full((B,), 38)
    targets = torch.tensor([[0, 1, 2]] * B)
    tl = torch.tensor(([3, 2, 1] * (B // 3 + 1))[:B])
    out = []
    for mode in ("ahead", "plain", "ahead-own-launch", "ahead-wrong-batch"):
        monkeypatch.delenv("HOWL_LSTM_RIDE_LOGMEL", raising=False)
        if mode == "ahead-own-launch":
            monkeypatch.setenv("HOWL_LSTM_RIDE_LOGMEL", "0")
        model = make("seq-lstm", C).train()
        tr = FusedTrainer(model, std, zmuv, lr=1e-3, weight_decay=1e-5)
        losses = []
        for i in range(3):
            nxt = None if mode == "plain" else batches[i + 1]
            if mode == "ahead-wrong-batch" and i == 0:
                nxt = batches[3]                 # announced, but step 1 trains on batches[1]: must be recomputed
            losses.append(tr.step_sequence(batches[i], lengths, targets, tl, 4, next_audio=nxt))
            if mode == "ahead" and i == 0:       # what the rider wrote for batch 1 == the frontend's own launch on batch 1
                assert tr._ahead is not None and tr._ahead[0] is batches[1]
                assert torch.equal(tr._ahead[1], std.log_mel_for_model(batches[1], zmuv))
        torch.cuda.synchronize()
        out.append([t.clone() for t in (tr.fp.flat, tr.m, tr.v, torch.stack([l.reshape(()) for l in losses]))])
    for other in out[1:]:
        for a, b in zip(out[0], other):
            assert torch.isfinite(a).all()
            assert torch.equal(a, b)


def test_seq_lstm_at_80_mel_bins_vs_oracle(monkeypatch):
    """NUM_MELS=80 exported (it configures the frontend, settings.py:32, AND the recurrent models' input width, rnn.py:36): the
    frontend's two filterbank banks, the input projection as its own GEMM (the fused projection is the 40-bin kernel), fused CTC
    step and the one-batch look-ahead (the next batch's frontend as its own launches behind the recurrence at this width)
    against the oracle."""
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    from howl_amd.model import RegisteredModel
    from howl_amd.settings import SETTINGS
    from howl_amd.training.fused import FusedTrainer
    from howl_amd.utils.synth import synthetic_pcm
    monkeypatch.setenv("NUM_MELS", "80")
    monkeypatch.setattr(SETTINGS.audio_transform, "num_mels", 80)
    B, L, C = 96, 8000, 5
    pcm = synthetic_pcm(B, L)
    std = StandardAudioTransform().to(DEV).eval()
    zmuv = ZmuvTransform().to(DEV)
    zmuv.update(std(pcm[:4].to(DEV)))
    sd0 = om.lstm_init(C, num_mels=80)
    model = RegisteredModel.find_registered_class("seq-lstm")(C)
    assert model.lstm.input_size == 80
    model.load_state_dict({k: v.clone() for k, v in sd0.items()})
    model = model.to(DEV).train()
    lengths = torch.sort(20 + torch.arange(B) % 19, descending=True).values
    targets = torch.tensor([[0, 1, 2]] * B)
    tl = torch.tensor([3] * B)
    tr = FusedTrainer(model, std, zmuv, lr=1e-3, weight_decay=1e-5)
    pcm2 = synthetic_pcm(B, L, seed=7)
    loss = tr.step_sequence(pcm.to(DEV), lengths, targets, tl, 4, next_audio=pcm2.to(DEV))
    grads = [g.clone() for g in tr.fp.grad_views]
    fb = ofe.mel_fb(80)
    z = ofe.Zmuv()
    z.update(ofe.standard_audio_transform(pcm[:4], fb))
    x_ref = z(ofe.standard_audio_transform(pcm, fb))
    sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
    ref, _ = om.seq_lstm_forward(sd, x_ref, lengths)
    ref_loss = torch.nn.CTCLoss(4)(torch.log_softmax(ref, -1), targets, lengths, tl)
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 1e-4
    for n, g in zip(om.lstm_param_names(), grads):       # (hot_parameters() order)
        r = sd[n].grad
        assert maxerr(g, r) < 1e-4 * max(1.0, r.abs().max().item()), n
    # the look-ahead features are the frontend's own
    assert tr._ahead is not None and torch.equal(tr._ahead[1], std.log_mel_for_model(pcm2.to(DEV), zmuv))
    loss2 = tr.step_sequence(pcm2.to(DEV), lengths, targets, tl, 4)
    assert torch.isfinite(loss2).all()


@pytest.mark.parametrize("B,T", [(512, 38), (257, 38), (64, 70), (1024, 20), (2049, 8), (300, 16), (129, 41), (40, 64), (2500, 1)])
def test_head_ctc_and_head_backward_rows_in_one_launch(B, T, monkeypatch):
    """Round 6 (howl_seq_head_ctc): between the two recurrences of the seq-lstm step, head forward + log_softmax / CTC + the head's
    backward over the rows as ONE launch in which a workgroup owns whole utterances (y1 stays in LDS), against the three launches
    it replaces (HOWL_SEQ_HEAD_FUSED=0): loss, logits, LSTM gradients and the first head layer's weight gradient bit for bit, the
    regrouped partial sums (dnn.0.bias, dnn.2.*) to rounding; BASELINE config 4's size, an odd batch, U = 1 (70 / 64 frames), several
    groups per workgroup (1024 x 20, 2049 x 8: an odd batch on more groups than CUs), groups of exactly one or two tiles (8, 16
    frames), 41 frames (82 rows: a sixth, two-row tile) and one-frame windows; ragged lengths, 0-3 labels; three steps each
    (AdamW in the fold) end in the same weights."""
    from howl_amd.training.fused import FusedTrainer
    rng = np.random.default_rng(B + T)
    feat = torch.from_numpy(rng.standard_normal((B, 1, 40, T)).astype(np.float32)).to(DEV)
    lengths = torch.sort(torch.from_numpy(rng.integers(max(min(4, T), T // 2), T + 1, B)), descending=True).values
    lengths[0] = T
    targets = torch.from_numpy(rng.integers(0, 4, (B, 3)))
    targets[::5, 1] = targets[::5, 0]                       # repeated labels
    tl = torch.minimum(torch.from_numpy(rng.integers(0, 4, B)), (lengths + 1) // 2)     # (alignable: repeats need a blank in between)
    out = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("HOWL_SEQ_HEAD_FUSED", fused)
        model = make("seq-lstm", 5).train()
        tr = FusedTrainer(model, None, None, lr=1e-3, weight_decay=1e-5)
        loss = tr.step_sequence_on_features(feat, lengths, targets, tl, 4)
        assert (model.ctc_nll is not None) == (fused == "1")
        first = (loss.clone(), tr.last_logits.clone(), [g.clone() for g in tr.fp.grad_views])
        for _ in range(2):
            tr.step_sequence_on_features(feat, lengths, targets, tl, 4)
        out[fused] = first + (tr.fp.flat.clone(),)
    a, b = out["1"], out["0"]
    assert torch.isfinite(a[0]) and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for n, ga, gb in zip(om.lstm_param_names(), a[2], b[2]):
        if n in ("dnn.0.bias", "dnn.2.weight", "dnn.2.bias"):
            assert maxerr(ga, gb) <= 2e-6 * max(1.0, gb.abs().max().item()), n
        else:
            assert torch.equal(ga, gb), n
    assert maxerr(a[3], b[3]) < 2e-5          # three AdamW steps (sign-like first steps amplify the regrouped sums' last bits)
