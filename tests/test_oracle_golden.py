"""Pin ``oracle/`` against golden vectors captured from the reference's own code (CPU, no GPU).

Fixtures: ``tests/golden/*.npz`` (made by ``tests/golden/make_golden.py``).  Tolerance: the oracle calls
the same ATen ops as the reference, so <= 1e-6 abs everywhere (mostly bit-exact); integer outputs exact.
"""
import numpy as np
import torch

from oracle import frontend as fe
from oracle import models as om

TOL = 1e-6


def t(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, tol=TOL):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a.double() - b.double()).abs().max().item() if a.numel() else 0.0
    assert err <= tol, err


def test_g1_filterbanks(golden):
    g = golden("g1_filterbanks")
    assert torch.equal(fe.mel_fb(40), t(g["fb_standard"]))
    for a in g["alphas"]:
        assert torch.equal(fe.mel_fb(40, alpha=float(a)), t(g[f"fb_vtlp_{a}"])), a
    # the alpha > 1 quirk (transform.py:397-401): corner points scaled then re-mapped
    assert abs(float(t(g["fb_vtlp_1.0999"]).sum()) - float(t(g["fb_standard"]).sum())) > 100


def test_g2_frontend_gsc(golden):
    g = golden("g2_frontend_gsc")
    audio = t(g["audio"])
    fb = fe.mel_fb(40)
    close(fe.standard_audio_transform(audio, fb), g["feats"])
    close(fe.standard_audio_transform(audio, fb, mels_only=True), g["mels"])
    close(fe.standard_audio_transform(audio, fe.mel_fb(40, alpha=float(g["vtlp_alpha"])), mels_only=True),
          g["mels_vtlp"])
    assert torch.equal(fe.compute_lengths(t(g["lens_in"])), t(g["lens_out"]))
    assert g["lens_out"].tolist() == [1, 1, 2, 38, 63, 78]


def test_g2_frontend_synth(golden):
    g = golden("g2_frontend_synth")
    fb = fe.mel_fb(40)
    for L in (8000, 16000, 13527):
        out = fe.standard_audio_transform(t(g[f"audio_{L}"]), fb, mels_only=True)
        assert out.shape[-1] == 1 + L // 200
        close(out, g[f"mels_{L}"])


def test_g4_zmuv(golden):
    g2, g4 = golden("g2_frontend_gsc"), golden("g4_zmuv")
    fb = fe.mel_fb(40)
    z = fe.Zmuv()
    audio, lengths = t(g2["audio"]), g2["lengths"]
    for i in np.argsort(-lengths, kind="stable"):  # generator fed the clips in file order, see below
        pass
    # make_golden feeds the six clips in file order, un-padded; recover them from the padded batch
    order = [0, 2, 3, 5, 4, 1]  # position of each file-order clip in the length-sorted batch
    for pos in order:
        L = int(lengths[pos])
        z.update(fe.standard_audio_transform(audio[pos:pos + 1, :L], fb))
    close(z.total, g4["total"], 0)
    close(z.mean, g4["mean"], 1e-6)
    close(z.mean2, g4["mean2"], 1e-5)
    z.mean, z.mean2 = t(g4["mean"]), t(g4["mean2"])
    close(z(fe.standard_audio_transform(audio, fb))[:, 0], g4["normed_ch0"], 1e-6)


def _features(golden):
    g2, g4 = golden("g2_frontend_gsc"), golden("g4_zmuv")
    z = fe.Zmuv()
    z.mean, z.mean2 = t(g4["mean"]), t(g4["mean2"])
    return z(t(g2["feats"])), z


def test_g5_res8(golden):
    x, _ = _features(golden)
    for C in (4, 12, 30):
        g = golden(f"g5_res8_c{C}")
        sd = om.res8_init(C)
        close(om.res8_forward(sd, x, False), g["eval_logits"], 1e-6)
        assert torch.equal(om.res8_forward(sd, x, False).argmax(1), t(g["eval_logits"]).argmax(1))
        names = om.res8_param_names()
        opt = om.AdamWState([sd[n] for n in names], 0.01, 1e-5)
        labels = t(g["labels"])
        for step in range(3):
            loss, logits, grads = om.train_step(lambda s, xx: om.res8_forward(s, xx, True), sd, names, opt, x, labels)
            close(loss, g[f"loss{step}"], 2e-6)
            if step == 0:
                close(logits, g["train_logits"], 2e-6)
                for n in names:
                    close(grads[n], g["grad0." + n], 2e-6)
                for i in (1, 6):
                    close(sd[f"bn{i}.running_mean"], g[f"bn{i}.running_mean.1"])
                    close(sd[f"bn{i}.running_var"], g[f"bn{i}.running_var.1"])
        for k, v in sd.items():
            close(v, g["sd3." + k], 2e-5)  # own AdamW write-out vs torch.optim.AdamW: op order differs
        close(om.res8_forward(sd, x, False), g["eval_logits_after3"], 5e-4)


def test_g5_res8_half_window(golden):
    g, g4 = golden("g5_res8_c4_half"), golden("g4_zmuv")
    z = fe.Zmuv()
    z.mean, z.mean2 = t(g4["mean"]), t(g4["mean2"])
    x = z(fe.standard_audio_transform(t(g["audio"]), fe.mel_fb(40)))
    assert x.shape[-1] == 41
    sd = om.res8_init(4)
    close(om.res8_forward(sd, x, False), g["eval_logits"], 1e-6)
    names = om.res8_param_names()
    opt = om.AdamWState([sd[n] for n in names], 0.01)
    loss, logits, grads = om.train_step(lambda s, xx: om.res8_forward(s, xx, True), sd, names, opt, x,
                                        torch.arange(6) % 4)
    close(logits, g["train_logits"], 2e-6)
    close(loss, g["loss0"], 2e-6)
    for n in ("conv0.weight", "conv1.weight", "conv6.weight", "output.weight"):
        close(grads[n], g["grad0." + n], 2e-6)


def test_g6_lstm(golden):
    x, _ = _features(golden)
    for name, fwd in (("lstm", om.lstm_forward), ("seq_lstm", om.seq_lstm_forward)):
        g = golden("g6_" + name)
        flen = t(g["frame_lengths"])
        assert flen.tolist() == sorted(flen.tolist(), reverse=True)
        for aten in (False, True):
            sd = om.lstm_init(5)
            logits, _ = fwd(sd, x, flen, aten=aten)
            close(logits, g["logits"], 2e-6)
        sd = {k: v.clone().requires_grad_(True) for k, v in om.lstm_init(5).items()}
        sc, _ = fwd(sd, x, flen)
        if name == "lstm":
            loss = torch.nn.functional.cross_entropy(sc, torch.arange(6) % 5)
        else:
            loss = torch.nn.CTCLoss(4)(torch.log_softmax(sc, -1), torch.tensor([[0, 1, 2]] * 6), flen,
                                       torch.tensor([3] * 6))
        loss.backward()
        close(loss.detach(), g["loss0"], 2e-6)
        for n in om.lstm_param_names():
            close(sd[n].grad, g["grad0." + n], 5e-6 * max(1.0, float(np.abs(g["grad0." + n]).max())))
        # streaming carry (rnn.py:62,67-68)
        sd = om.lstm_init(5)
        if name == "lstm":
            a, hc = fwd(sd, x[:1, :, :, :40], torch.tensor([40]))
            b, _ = fwd(sd, x[:1, :, :, 40:], torch.tensor([41]), hx=hc)
        else:
            a, hc = fwd(sd, x[:1, :, :, :40], None)
            b, _ = fwd(sd, x[:1, :, :, 40:], None, hx=hc)
        close(a, g["stream_a"], 2e-6)
        close(b, g["stream_b"], 2e-6)


def test_g15_whole_clips(golden):
    """The recurrent oracles at whole-clip lengths (318 / 258 / 206 / 128 frames) against the reference's classes: the sequence
    objective batches clips, not windows (batchifier.py:14-34, train.py:198-200,291-296)."""
    for name, fwd in (("lstm", om.lstm_forward), ("seq_lstm", om.seq_lstm_forward)):
        g = golden("g15_whole_clips_" + name)
        x, flen = t(g["x"]), t(g["frame_lengths"])
        assert flen.tolist() == [318, 258, 206, 128]
        logits, _ = fwd(om.lstm_init(5), x, flen)
        close(logits, g["logits"], 5e-6)
        sd = {k: v.clone().requires_grad_(True) for k, v in om.lstm_init(5).items()}
        sc, _ = fwd(sd, x, flen)
        if name == "lstm":
            loss = torch.nn.functional.cross_entropy(sc, torch.arange(4) % 5)
        else:
            loss = torch.nn.CTCLoss(4)(torch.log_softmax(sc, -1), t(g["targets"]), flen, t(g["target_lengths"]))
        loss.backward()
        close(loss.detach(), g["loss0"], 5e-6 * max(1.0, float(g["loss0"])))
        for n in om.lstm_param_names():
            close(sd[n].grad, g["grad0." + n], 4e-5 * max(1.0, float(np.abs(g["grad0." + n]).max())))   # BPTT over 318 steps in fp32
        sd = om.lstm_init(5)
        if name == "lstm":
            a, hc = fwd(sd, x[:1, :, :, :160], torch.tensor([160]))
            b, _ = fwd(sd, x[:1, :, :, 160:], torch.tensor([161]), hx=hc)
        else:
            a, hc = fwd(sd, x[:1, :, :, :160], None)
            b, _ = fwd(sd, x[:1, :, :, 160:], None, hx=hc)
        close(a, g["stream_a"], 5e-6)
        close(b, g["stream_b"], 5e-6)
    # the frontend oracle on the same clips (the seq-lstm file carries the audio)
    g = golden("g15_whole_clips_seq_lstm")
    z = _features(golden)[1]
    feats = z(fe.standard_audio_transform(t(g["audio"]), fe.mel_fb(40)))
    close(feats[:, :1], g["x"], 2e-4)


def test_g7_specaug(golden):
    g = golden("g7_specaug")
    out = fe.spec_augment_apply(t(g["x"]), g["f0"], g["f"], g["t0"], g["t"])
    assert torch.equal(out, t(g["out"]))


def _g9_inputs(g):
    wfs = [torch.arange(int(L), dtype=torch.float32) * 1e-5 for L in g["wf_lens"]]
    bgs = [torch.full((int(L),), 0.1 * (i + 1)) + torch.arange(int(L), dtype=torch.float32) * 1e-6
           for i, L in enumerate(g["bg_lens"])]
    return wfs, bgs


def _g9_sub(a):
    a = np.asarray(a, np.float32)
    return np.concatenate([a[:8], a[8:-8:61], a[-8:]])


def test_g9_dataset_mixer(golden):
    """DatasetMixer restatement vs the reference class's own outputs (same `random` stream)."""
    import random
    g = golden("g9_mixer")
    wfs, bgs = _g9_inputs(g)
    for trial, seed in enumerate((0, 3, 11)):
        rand = random.Random(seed)
        out = fe.dataset_mixer(rand, wfs, bgs)
        for i, o in enumerate(out):
            np.testing.assert_allclose(_g9_sub(o.numpy()), g[f"mixed_{trial}_{i}"], rtol=0, atol=1e-7)
        assert rand.random() == float(g[f"next_draw_{trial}"])


def g10_inputs(g):
    """Clips and label maps of golden G10 (ramps whose values name (clip, offset); maps stored in CSR form)."""
    clips = [torch.arange(int(L), dtype=torch.float32) * 1e-6 + 0.05 * (i + 1) for i, L in enumerate(g["clip_lens"])]
    ptr = g["map_ptr"]
    maps = [{float(k): int(v) for k, v in zip(g["map_end_ms"][a:b], g["map_label"][a:b])} for a, b in zip(ptr[:-1], ptr[1:])]
    return clips, maps


G10_VARIANTS = [(0, {}), (7, {}), (21, dict(positive_sample_prob=0.8, window_size_ms=1000)),
                (3, dict(pad_to_window=False, window_size_ms=250))]


def g10_check(g, trial, audio, labels, lengths):
    a = np.asarray(audio, np.float32)
    assert a.shape[1] == int(g[f"width_{trial}"])
    assert np.array_equal(np.asarray(labels), g[f"labels_{trial}"])
    assert np.array_equal(np.asarray(lengths), g[f"lengths_{trial}"])
    nz = a != 0
    assert np.array_equal(nz.sum(1), g[f"nz_count_{trial}"])
    assert np.array_equal([int(r.argmax()) if r.any() else -1 for r in nz], g[f"nz_start_{trial}"])
    assert np.array_equal(a[:, ::53], g[f"every53_{trial}"])                    # bit-exact: pure data movement
    assert np.array_equal([r[m][0] if m.any() else 0.0 for r, m in zip(a, nz)], g[f"first_{trial}"])
    assert np.array_equal([r[m][-1] if m.any() else 0.0 for r, m in zip(a, nz)], g[f"last_{trial}"])


def test_g10_frame_batchifier(golden):
    """WakeWordFrameBatchifier restatement vs the reference class's own batches under the same `random` stream."""
    import random
    from oracle import collate as oc
    g = golden("g10_frame_batchifier")
    clips, maps = g10_inputs(g)
    for trial, (seed, kw) in enumerate(G10_VARIANTS):
        rand = random.Random(seed)
        audio, labels, lengths = oc.frame_batchify(rand, clips, maps, 4, **kw)
        g10_check(g, trial, audio.numpy(), labels.numpy(), lengths.numpy())
        assert rand.random() == float(g[f"next_draw_{trial}"])


def test_g7b_collate_chain(golden):
    """truncate -> Timeshift -> Noise restatement: consumes the reference's draws and produces its crops."""
    import random
    from oracle import collate as oc
    g = golden("g7b_collate_protocol")
    lens = [int(v) for v in g["lens"]]
    for trial, seed in enumerate((0, 1, 2, 5)):
        rand = random.Random(seed)
        draws = []

        class Rec:
            def random(self):
                v = rand.random()
                draws.append(v)
                return v

        torch.manual_seed(seed)
        clips = oc.truncate_length([torch.arange(L, dtype=torch.float32) * 1e-5 for L in lens], 16000)
        out = oc.noise(Rec(), oc.timeshift(Rec(), clips))
        assert np.array_equal(np.array(draws), g[f"draws_{trial}"])
        assert [o.numel() for o in out] == g[f"out_len_{trial}"].tolist()
        # torch's CPU generator is the reference's too: same seed, same noise samples, bit-identical first samples
        assert np.array_equal(np.array([float(o[0]) for o in out]), g[f"first_{trial}"])


def test_g13_stock_80_mel_bins(golden):
    """G13: the reference at its stock NUM_MELS = 80 (settings.py:32) -- filterbanks, frontend, ZMUV, res8 eval / one training step."""
    g = golden("g13_res8_80mel")
    assert torch.equal(fe.mel_fb(80), t(g["fb_standard"]))
    for a in ("0.9", "1.0999"):
        assert torch.equal(fe.mel_fb(80, alpha=float(a)), t(g[f"fb_vtlp_{a}"])), a
    audio = t(g["audio"])
    fb = fe.mel_fb(80)
    feats = fe.standard_audio_transform(audio, fb)
    close(feats, g["feats"])
    close(fe.standard_audio_transform(audio, fe.mel_fb(80, alpha=float(g["vtlp_alpha"])), mels_only=True), g["mels_vtlp"])
    z = fe.Zmuv()
    z.mean, z.mean2 = t(g["zmuv_mean"]), t(g["zmuv_mean2"])
    x = z(t(g["feats"]))
    C = 12
    sd = om.res8_init(C)
    close(om.res8_forward(sd, x, False), g["eval_logits"], 1e-6)
    names = om.res8_param_names()
    opt = om.AdamWState([sd[n] for n in names], 0.01, 1e-5)
    loss, logits, grads = om.train_step(lambda s, xx: om.res8_forward(s, xx, True), sd, names, opt, x, t(g["labels"]))
    close(loss, g["loss0"], 2e-6)
    close(logits, g["train_logits"], 2e-6)
    for n in names:
        close(grads[n], g["grad0." + n], 2e-6)
    for i in (1, 6):
        close(sd[f"bn{i}.running_mean"], g[f"bn{i}.running_mean.1"])
        close(sd[f"bn{i}.running_var"], g[f"bn{i}.running_var.1"])
    close(om.res8_forward(sd, x, False), g["eval_logits_after1"], 2e-4)


def test_g14_two_second_windows(golden):
    """G14: the reference's res8 on 161-frame inputs (2-s windows) at 40 and 80 mel bins, eval and one training step."""
    g = golden("g14_res8_two_second_windows")
    names = om.res8_param_names()
    for mels in (40, 80):
        pre = f"m{mels}."
        x = t(g[pre + "x"])
        assert x.shape == (3, 1, mels, 161)
        sd = om.res8_init(12)
        close(om.res8_forward(sd, x, False), g[pre + "eval_logits"], 1e-6)
        opt = om.AdamWState([sd[n] for n in names], 0.01, 1e-5)
        loss, logits, grads = om.train_step(lambda s, xx: om.res8_forward(s, xx, True), sd, names, opt, x, torch.arange(3) % 12)
        close(loss, g[pre + "loss0"], 2e-6)
        close(logits, g[pre + "train_logits"], 2e-6)
        for n in names:
            close(grads[n], g[pre + "grad0." + n], 2e-6)
        for i in (1, 6):
            close(sd[f"bn{i}.running_mean"], g[pre + f"bn{i}.running_mean.1"])
            close(sd[f"bn{i}.running_var"], g[pre + f"bn{i}.running_var.1"])
