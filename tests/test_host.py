"""CPU tests of the host-side logic (settings, registry, RNG protocol helpers, no-fallback guards)."""
import importlib
import os

import pytest
import numpy as np
import torch


def test_settings_env(monkeypatch):
    from howl_amd import settings
    monkeypatch.setenv("NUM_MELS", "40")
    monkeypatch.setenv("VOCAB", '["hey","fire","fox"]')
    monkeypatch.setenv("LEARNING_RATE", "0.01")
    monkeypatch.setenv("USE_NOISE_DATASET", "False")
    monkeypatch.setenv("INFERENCE_SEQUENCE", "[0,1,2]")
    s = settings.HowlSettings()
    assert s.audio_transform.num_mels == 40 and s.audio_transform.hop_length == 200
    assert s.training.vocab == ["hey", "fire", "fox"] and s.training.learning_rate == 0.01
    assert s.training.use_noise_dataset is False and s.inference_engine.inference_sequence == [0, 1, 2]
    monkeypatch.delenv("NUM_MELS")
    s.reset()
    assert s.audio_transform.num_mels == 80   # the reference's default (settings.py:32)


def test_registry_and_state_dict_keys():
    from howl_amd.model import RegisteredModel
    assert "res8" in RegisteredModel.registered_names()
    with pytest.raises(KeyError):
        RegisteredModel.find_registered_class("nope")
    m = RegisteredModel.find_registered_class("res8")(12)
    keys = list(m.state_dict().keys())
    assert keys[:5] == ["conv0.weight", "bn1.running_mean", "bn1.running_var", "bn1.num_batches_tracked", "conv1.weight"]
    assert keys[-2:] == ["output.weight", "output.bias"] and len(keys) == 27
    assert sum(p.numel() for p in m.parameters()) == 110307
    assert m.streaming().is_streaming and not m.static().is_streaming and m.compute_length(7) == 7


def test_same_seed_same_init_as_reference_order():
    """Construction order matches cnn.py:116-125, so a seed gives the weights a stock nn build would get."""
    import torch.nn as nn
    from howl_amd.model import RegisteredModel
    torch.manual_seed(0)
    m = RegisteredModel.find_registered_class("res8")(12)
    torch.manual_seed(0)
    conv0 = nn.Conv2d(1, 45, (3, 3), padding=(1, 1), bias=False)
    convs = [nn.Conv2d(45, 45, (3, 3), padding=1, bias=False) for _ in range(6)]
    out = nn.Linear(45, 12)
    assert torch.equal(m.conv0.weight, conv0.weight) and torch.equal(m.conv6.weight, convs[5].weight)
    assert torch.equal(m.output.weight, out.weight)


def test_closed_form_matches_oracle_copy():
    from howl_amd.utils.synth import res8_closed_form_state
    from oracle import models as om
    a, b = res8_closed_form_state(12), om.res8_init(12)
    for k, v in a.items():
        assert torch.equal(v, b[k]), k


def test_vtlp_points_reproduce_reference_filterbank(golden):
    """Host-side corner-point warp + the triangle formula == the reference's create_vtlp_fb_matrix goldens."""
    from howl_amd.data.transform.transform import mel_corner_points, vtlp_warp_points
    g = golden("g1_filterbanks")
    all_freqs = torch.linspace(0, 8000, 257)
    for a in g["alphas"]:
        pts = vtlp_warp_points(mel_corner_points(40, 16000), float(a), 16000)
        f_diff = pts[1:] - pts[:-1]
        slopes = pts.unsqueeze(0) - all_freqs.unsqueeze(1)
        fb = torch.max(torch.zeros(1), torch.min((-1.0 * slopes[:, :-2]) / f_diff[:-1], slopes[:, 2:] / f_diff[1:]))
        assert torch.equal(fb, torch.from_numpy(g[f"fb_vtlp_{a}"])), a


def test_no_cpu_fallback():
    from howl_amd import lib, ops
    with pytest.raises(lib.HowlHipError):
        ops._p(torch.zeros(4))


def test_device_collate_follows_reference_rng_protocol(golden):
    """DeviceCollate.draw consumes random.random() in exactly the order/roles of the reference chain
    truncate -> Timeshift.train() -> Noise.train() (G7b captured from the reference's modules)."""
    from howl_amd.data.collate import DeviceCollate
    g = golden("g7b_collate_protocol")
    lens = g["lens"].tolist()

    class Replay:
        def __init__(self, values):
            self.values, self.i = list(values), 0

        def random(self):
            v = self.values[self.i]
            self.i += 1
            return v

    for trial in range(4):
        dc = DeviceCollate(torch.zeros(1, 1), torch.tensor(lens), torch.zeros(len(lens), dtype=torch.long), 16000)
        dc.rand = Replay(g[f"draws_{trial}"])
        tl, shift, head, sigma, sp = dc.draw(list(range(len(lens))))
        assert dc.rand.i == len(g[f"draws_{trial}"]), "consumed a different number of draws than the reference"
        assert [l - w for l, w in zip(tl, shift)] == g[f"out_len_{trial}"].tolist()
        for k in range(len(lens)):          # ramp input: first sample ~ crop offset * 1e-5 (noise is <= ~4e-3)
            off = shift[k] if head[k] else 0
            assert abs(g[f"first_{trial}"][k] - off * 1e-5) < 6e-3
            if shift[k] > 600 and head[k]:
                assert g[f"first_{trial}"][k] > 0.5 * shift[k] * 1e-5 - 5e-3


def test_device_collate_array_draws_are_the_reference_draws(golden):
    """The batched form the entry points use (``DeviceCollate.draw_arrays``: numpy's MT19937 loaded with the ``random``
    stream's state) against G7b -- the golden's trials were drawn from ``random.seed(seed)``, so a ``random.Random(seed)`` must
    give the same crops and end at the same position of the stream -- and against the scalar ``draw`` element for element, for
    a private stream (kept in numpy between batches) and for the global ``random`` module (handed back after every batch)."""
    import random
    from howl_amd.data.collate import DeviceCollate
    g = golden("g7b_collate_protocol")
    lens = g["lens"].tolist()
    mk = lambda ls, **kw: DeviceCollate(torch.zeros(1, 1), torch.tensor(ls), torch.zeros(len(ls), dtype=torch.long), 16000, **kw)
    for trial, seed in enumerate((0, 1, 2, 5)):
        dc = mk(lens)
        dc.rand = random.Random(seed)
        tl, shift, head, sigma, sp = dc.draw_arrays(list(range(len(lens))))
        assert (tl - shift).tolist() == g[f"out_len_{trial}"].tolist()
        probe = random.Random(seed)
        for v in g[f"draws_{trial}"]:
            assert probe.random() == float(v)
        assert dc.rand.random() == probe.random(), "consumed a different number of draws than the reference"
    rng = np.random.default_rng(3)
    many = rng.integers(6000, 20000, size=700).tolist()
    for seed in (0, 1, 7, None):
        if seed is None:
            random.seed(1234)
            st = random.getstate()
        a, b = mk(many, seed=seed), mk(many, seed=seed)
        got, want = [], []
        for rep in range(4):
            ids = rng.permutation(len(many))[:256]
            want.append((ids, a.draw(ids.tolist())))
        tail = a.rand.random()
        if seed is None:
            random.setstate(st)
        for ids, w in want:
            r = b.draw_arrays(ids)
            for x, y in zip(w, r):
                assert x == [type(x[0])(v) for v in y.tolist()]
        assert b.rand.random() == tail
        if seed is not None:      # scalar draws after array draws see the advanced stream
            ids = rng.permutation(len(many))[:64].tolist()
            assert b.draw(ids) == a.draw(ids)


def test_device_collate_mixer_draws_match_reference(golden):
    """DeviceCollate.draw_mixer makes DatasetMixer's draws (background choice, window, alpha) in the reference's order: the
    mix rebuilt from them equals the reference class's output (G9) and the random stream ends at the same position."""
    import random
    from howl_amd.data.collate import DeviceCollate
    g = golden("g9_mixer")
    wf_lens, bg_lens = g["wf_lens"].tolist(), g["bg_lens"].tolist()
    for trial, seed in enumerate((0, 3, 11)):
        dc = DeviceCollate(torch.zeros(1, 1), torch.tensor(wf_lens), torch.zeros(len(wf_lens), dtype=torch.long), 1 << 20,
                           background=(torch.zeros(1, 1), bg_lens))
        dc.rand = random.Random(seed)
        bg_id, bg_off, alpha = dc.draw_mixer(wf_lens)
        assert dc.rand.random() == float(g[f"next_draw_{trial}"])
        for i, L in enumerate(wf_lens):
            n = np.arange(L, dtype=np.float32)
            wf = n * np.float32(1e-5)
            bg = np.float32(0.1 * (bg_id[i] + 1)) + (n + np.float32(bg_off[i])) * np.float32(1e-6)
            mixed = wf * np.float32(1 - alpha[i]) + bg * np.float32(alpha[i]) if alpha[i] else wf
            sub = np.concatenate([mixed[:8], mixed[8:-8:61], mixed[-8:]])
            np.testing.assert_allclose(sub, g[f"mixed_{trial}_{i}"], rtol=0, atol=2e-6)


def test_export_honkling_wire_format(tmp_path):
    """export_honkling.py:19-35: `weights['RES8'] = {json of the state_dict + unit scale vectors}` with the reference's keys."""
    import json
    from howl_amd.training.run import export_honkling
    from oracle import models as om
    sd = om.res8_init(12)
    src, dst = tmp_path / "model-best.pt.bin", tmp_path / "w.js"
    torch.save(sd, src)
    export_honkling.main(["-i", str(src), "-o", str(dst), "--name", "RES8"])
    text = dst.read_text()
    prefix = "weights['RES8'] = "
    assert text.startswith(prefix)
    d = json.loads(text[len(prefix):])
    assert set(d) == set(sd) | {"scale1.scale", "scale3.scale", "scale5.scale"}
    assert d["scale3.scale"] == [1.0] * 45
    assert np.allclose(np.array(d["conv3.weight"], np.float32), sd["conv3.weight"].numpy())


def test_converted_static_model_windows():
    """ConvertedStaticModel (base.py:40-62): window sequence incl. the first-window quirk, against a direct restatement of
    the reference loop (first window = x[..., win:], then complete strided windows)."""
    from howl_amd.model.base import ConvertedStaticModel

    class Probe(torch.nn.Module):
        num_labels = 3

        def forward(self, w, lengths):
            return torch.stack([torch.tensor(float(w.size(-1))), w.sum(), w.abs().sum()])   # empty first window is legal

    def reference_loop(model, x, win, hop):
        first, window, idx, outs = True, x[:, :, :, win:], hop, []
        while first or window.size(3) == win:
            first = False
            outs.append(model(window, None))
            window = x[:, :, :, idx: idx + win]
            idx += hop
        return torch.stack(outs)

    for T, win, hop in [(100, 20, 10), (55, 20, 7), (20, 20, 5), (19, 20, 5), (41, 10, 10)]:
        x = torch.randn(2, 1, 4, T)
        m = ConvertedStaticModel(Probe(), win, hop)
        got, want = m(x, None), reference_loop(Probe(), x, win, hop)
        assert got.shape == want.shape and torch.allclose(got, want), (T, win, hop)
        assert m.compute_length(T) == max(1, (T - win) // hop) and m.compute_length(None) is None


def test_phone_token_context_vs_reference_golden(tmp_path):
    """InferenceContext(token_type="phone") with PhoneticFrameLabeler / PhoneticTranscriptSearcher / LabelColoring / PhonePhrase
    against golden G11 -- the reference's own classes on the same small pronunciation dictionary
    (tests/golden/make_golden.py::phone_context_golden)."""
    import json
    from pathlib import Path
    from types import SimpleNamespace
    from howl_amd.context import InferenceContext
    from howl_amd.data.common.phone import PhonePhrase, PronunciationDictionary
    from howl_amd.settings import SETTINGS
    g = json.loads((Path(__file__).parent / "golden" / "g11_phone_context.json").read_text())
    inp = g["inputs"]
    dpath = tmp_path / "test.dict"
    dpath.write_text(inp["dictionary"])
    saved = (SETTINGS.training.phone_dictionary, SETTINGS.inference_engine.inference_sequence)
    SETTINGS.training.phone_dictionary = str(dpath)
    SETTINGS.inference_engine.inference_sequence = [0, 1, 2]
    try:
        d = PronunciationDictionary.from_file(dpath)
        assert "HEY" in d and " fox " in d and "xyzzy" not in d and len(d.encode("hey")) == 1 and str(d.encode("it's")[0]) == "ih1 t s"
        with pytest.raises(ValueError):
            d.encode("xyzzy")
        for tag, use_blank in (("noblank", False), ("blank", True)):
            ctx = InferenceContext(inp["vocab"], token_type="phone", use_blank=use_blank)
            want = g[tag]
            assert list(ctx.adjusted_vocab) == want["adjusted_vocab"]
            assert (ctx.num_labels, ctx.negative_label, ctx.blank_label) == (want["num_labels"], want["negative_label"], want["blank_label"])
            assert {str(k): v for k, v in ctx.coloring.color_map.items()} == want["color_map"]
            assert ctx.searcher.pattern.pattern == want["pattern"]
        for tr, want in zip(inp["transcripts"], g["frame_labels"]):
            md = SimpleNamespace(transcription=tr, end_timestamps=[10.0 * (i + 1) for i in range(len(tr))])
            got = ctx.labeler.compute_frame_labels(md).timestamp_label_map
            assert {str(k): v for k, v in got.items()} == want, tr
        assert [ctx.searcher.search(q) for q in inp["queries"]] == g["search"]
        assert [ctx.searcher.contains_any(q) for q in inp["queries"]] == g["contains_any"]
        pp = PhonePhrase.from_string("hh ey1 sil f ay1 sp er0 spn")
        ph = g["phrase"]
        assert pp.audible_transcript == ph["audible"] and pp.sil_indices == ph["sil"]
        assert [pp.all_idx_to_transcript_idx(i) for i in range(len(pp.phones))] == ph["all_to_transcript"]
        assert [pp.audible_idx_to_all_idx(i) for i in range(len(pp.audible_phones))] == ph["audible_to_all"]
        assert pp.audible_index(PhonePhrase.from_string("sil er0")) == ph["index_er0"]
        assert pp.audible_index(PhonePhrase.from_string("ay1 er0"), 1) == ph["index_from1"]
        with pytest.raises(ValueError):
            pp.audible_index(PhonePhrase.from_string("sil"))
        with pytest.raises(ValueError):
            pp.all_idx_to_transcript_idx(len(pp.phones))
        for w, want in g["transform"].items():
            assert str(ctx.labeler.transform(w)) == want, w
        with pytest.raises(ValueError):
            ctx.labeler.transform("<unk>")           # the reference's loop resumes at '>' and fails (labeler.py:77-89)
        with pytest.raises(ValueError):
            InferenceContext(inp["vocab"], token_type="syllable")
    finally:
        SETTINGS.training.phone_dictionary, SETTINGS.inference_engine.inference_sequence = saved


def test_reference_written_workspace_round_trip(golden, tmp_path):
    """f4 against a file the REFERENCE wrote (tests/golden/ref_workspace: ``howl.workspace.Workspace.save_model`` and the
    ``zmuv.pt.bin`` of ``pretrain_gsc.py:106``, captured by make_golden.py --only-checkpoint): the product's ``Workspace``
    loads it into the product's ``Res8`` / ``ZmuvTransform`` with no key left over, the loaded weights reproduce the
    reference's eval logits through the oracle, and a workspace the PRODUCT writes holds the same keys / shapes / dtypes (so
    stock Howl's ``load_state_dict`` accepts it, ``hubconf.py:53-84``)."""
    import shutil
    from pathlib import Path
    import numpy as np
    import torch
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.model import RegisteredModel
    from howl_amd.workspace import Workspace
    from oracle import frontend as ofe, models as om
    g = golden("g12_ref_workspace")
    src = Path(__file__).resolve().parent / "golden" / "ref_workspace"
    ws = Workspace(tmp_path / "ws", delete_existing=False)
    for name in ("model-best.pt.bin", "zmuv.pt.bin"):
        shutil.copyfile(src / name, ws.path / name)
    C = int(g["num_labels"])
    model = RegisteredModel.find_registered_class("res8")(C)
    ws.load_model(model, best=True)                                   # strict load: every key of the reference's file is consumed
    zmuv = ZmuvTransform()
    zmuv.load_state_dict(torch.load(str(ws.path / "zmuv.pt.bin"), map_location="cpu"))
    sd = model.state_dict()
    assert list(sd.keys()) == list(g["keys"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g["shapes"])
    assert [str(v.dtype) for v in sd.values()] == list(g["dtypes"])
    assert list(zmuv.state_dict().keys()) == list(g["zmuv_keys"])
    # the loaded numbers are the reference's: its eval logits come back through the oracle (pinned elsewhere) to rounding
    z = ofe.Zmuv()
    z.total, z.mean, z.mean2 = zmuv.total.clone(), zmuv.mean.clone(), zmuv.mean2.clone()
    x = z(ofe.standard_audio_transform(torch.from_numpy(g["audio"]), ofe.mel_fb(40)))
    with torch.no_grad():
        logits = om.res8_forward({k: v.clone() for k, v in sd.items()}, x, False)
    assert np.abs(logits.numpy() - g["eval_logits"]).max() < 1e-5
    # the reverse direction: a product-written workspace is a bare state_dict with the reference's keys, shapes and dtypes
    ws2 = Workspace(tmp_path / "ws2", delete_existing=False)
    ws2.save_model(model, best=True)
    torch.save({k: v.cpu() for k, v in zmuv.state_dict().items()}, str(ws2.path / "zmuv.pt.bin"))
    ref_file = torch.load(str(src / "model-best.pt.bin"), map_location="cpu")
    own_file = torch.load(ws2.model_path(best=True), map_location="cpu")
    assert type(own_file) is type(ref_file) or isinstance(own_file, dict)
    assert list(own_file.keys()) == list(ref_file.keys())
    for k in ref_file:
        assert own_file[k].dtype == ref_file[k].dtype and own_file[k].shape == ref_file[k].shape and torch.equal(own_file[k], ref_file[k]), k
    zr, zo = torch.load(str(src / "zmuv.pt.bin")), torch.load(str(ws2.path / "zmuv.pt.bin"))
    assert list(zr.keys()) == list(zo.keys()) and all(torch.equal(zr[k], zo[k]) and zr[k].dtype == zo[k].dtype for k in zr)


def test_res8_warns_about_other_mel_counts_and_the_entry_points_refuse(monkeypatch, caplog):
    """The res8 kernels take 40 mel bins (every res8 preset) and 80 (stock Howl's default, settings.py:32).  With any other count,
    constructing the model still works (to load / convert / inspect a state_dict, as ``cnn.py:113`` allows) and says so; the entry
    points, which build the frontend from the same settings, fail before the first batch and name the environment variable."""
    import logging
    import pytest
    from howl_amd.model import RegisteredModel
    from howl_amd.model.cnn import require_supported_mels
    from howl_amd.settings import SETTINGS
    monkeypatch.setattr(SETTINGS.audio_transform, "num_mels", 64)
    with caplog.at_level(logging.WARNING):
        model = RegisteredModel.find_registered_class("res8")(12)
    assert "NUM_MELS=40" in caplog.text
    assert sorted(model.state_dict())[:2] == ["bn1.num_batches_tracked", "bn1.running_mean"]
    with pytest.raises(ValueError, match="NUM_MELS=40"):
        require_supported_mels(model)
    for ok in (40, 80):
        monkeypatch.setattr(SETTINGS.audio_transform, "num_mels", ok)
        caplog.clear()
        with caplog.at_level(logging.WARNING):
            RegisteredModel.find_registered_class("res8")(12)
        assert "NUM_MELS" not in caplog.text
        require_supported_mels(model)


def test_vtlp_corner_points_numpy_path_is_bit_identical():
    """The frontend's per-step VTLP warp runs on numpy float32 (``vtlp_warp_points_np``: 8 us) instead of eight torch ops on a
    42-element tensor (100 us of host time on 75 % of the training steps); both are restatements of transform.py:394-401 and
    must give the same bits for every alpha the reference can draw (``random.random() * 0.2 + 0.9``), 40 / 80 / odd mel counts,
    the alpha > 1 re-masking quirk included."""
    import random
    import numpy as np
    from howl_amd.data.transform.transform import mel_corner_points, vtlp_warp_points, vtlp_warp_points_np
    rnd = random.Random(7)
    for mels in (40, 80, 13):
        pts = mel_corner_points(mels, 16000)
        p32 = pts.numpy().copy()
        alphas = [0.9, 1.0, 1.0999999, 1.1, 0.9000001] + [rnd.random() * 0.2 + 0.9 for _ in range(3000)]
        for alpha in alphas:
            a = vtlp_warp_points(pts, alpha, 16000).numpy()
            b = vtlp_warp_points_np(p32, alpha, 16000)
            assert b.dtype == np.float32 and np.array_equal(a.view(np.uint32), b.view(np.uint32)), (mels, alpha)
        assert np.array_equal(p32, pts.numpy())      # inputs untouched
