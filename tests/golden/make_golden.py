"""Capture golden vectors from the REFERENCE's own code (run by hand; needs /root/reference).

    NUM_MELS=40 python tests/golden/make_golden.py

Imports castorini/howl from /root/reference under the shims in ``_reference_shims.py`` and records
inputs + outputs of the hot path as small ``.npz`` fixtures next to this file.  The fixtures are data
(inputs and expected outputs); no reference source travels.  ``tests/test_oracle_golden.py`` pins
``oracle/`` against them; the ``-m gpu`` tests pin the HIP path against them as well.

Inputs: the six 16 kHz GSC clips the reference's tests hold
(``test/test_data/datasets/google-speech-commands/{cat,dog}/*.wav``, decoded int16/32768 like
soundfile does) plus closed-form synthetic clips.  Model weights are closed-form (no RNG, nothing to
commit): ``oracle.models.res8_init`` / ``lstm_init``.
"""
import os
import random
import sys
import wave
from pathlib import Path

import numpy as np
import torch

os.environ.setdefault("NUM_MELS", "40")
HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent.parent))
import _reference_shims  # noqa: E402

_reference_shims.install()

from howl.context import InferenceContext  # noqa: E402
from howl.data.transform.operator import ZmuvTransform  # noqa: E402
from howl.data.transform.transform import (SpecAugmentTransform, StandardAudioTransform,  # noqa: E402
                                           create_vtlp_fb_matrix)
from howl.model import RegisteredModel  # noqa: E402
from howl.model.inference import FrameInferenceEngine, InferenceEngine  # noqa: E402
from howl.settings import SETTINGS  # noqa: E402
from howl.utils.audio_utils import stride  # noqa: E402

from oracle import models as om  # noqa: E402  (closed-form weights only)

torch.manual_seed(0)
GSC = Path(_reference_shims.REFERENCE) / "test/test_data/datasets/google-speech-commands"
WAVS = ["cat/0ab3b47d_nohash_0.wav", "cat/0ab3b47d_nohash_1.wav", "cat/0ac15fe9_nohash_0.wav",
        "dog/0a7c2a8d_nohash_0.wav", "dog/0ab3b47d_nohash_0.wav", "dog/0ac15fe9_nohash_0.wav"]


def read_wav(path):
    with wave.open(str(path), "rb") as w:
        assert w.getframerate() == 16000 and w.getnchannels() == 1 and w.getsampwidth() == 2
        return np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").astype(np.float32) / 32768.0


def batchify_like_reference(clips):
    """operator.py:77-86: sort by length descending, zero-pad right to the longest."""
    clips = sorted(clips, key=lambda c: -len(c))
    lengths = torch.tensor([len(c) for c in clips])
    audio = torch.zeros(len(clips), max(len(c) for c in clips))
    for i, c in enumerate(clips):
        audio[i, : len(c)] = torch.from_numpy(c)
    return audio, lengths


def synthetic_clips(L):
    n = np.arange(L, dtype=np.float64)
    out = []
    lcg = 12345
    noise = np.empty(L)
    for i in range(L):
        lcg = (1103515245 * lcg + 12345) % (1 << 31)
        noise[i] = lcg / float(1 << 30) - 1.0
    out.append((0.3 * np.sin(2 * np.pi * 440.0 * n / 16000 + 0.5) + 0.05 * noise).astype(np.float32))
    out.append(np.zeros(L, np.float32))
    out.append(np.where((n // 40) % 2 == 0, 1.0, -1.0).astype(np.float32))
    imp0 = np.zeros(L, np.float32); imp0[0] = 1.0
    impl = np.zeros(L, np.float32); impl[L - 1] = 1.0
    out += [imp0, impl]
    return np.stack(out)


def save(name, **arrays):
    arrays = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()}
    np.savez_compressed(HERE / f"{name}.npz", **arrays)
    print(f"{name}.npz: " + ", ".join(f"{k}{list(v.shape)}" for k, v in arrays.items()))


def main():
    clips = [read_wav(GSC / w) for w in WAVS]
    audio, lengths = batchify_like_reference(clips)
    std = StandardAudioTransform().eval()

    # ---- G1: filterbanks -------------------------------------------------------------------------
    alphas = [0.9, 0.95, 1.0, 1.05, 1.0999]
    fbs = {"fb_standard": create_vtlp_fb_matrix(257, 0.0, 8000.0, 40, 16000, 1.0, training=False)}
    for a in alphas:
        fbs[f"fb_vtlp_{a}"] = create_vtlp_fb_matrix(257, 0.0, 8000.0, 40, 16000, a, training=True)
    save("g1_filterbanks", alphas=np.array(alphas), **fbs)

    # ---- G2/G3: frontend ---------------------------------------------------------------------------
    feats = std(audio)
    mels = std(audio, mels_only=True)
    lens_in = torch.tensor([512, 711, 712, 8000, 12971, 16000])
    std.train()
    random.seed(11)
    assert random.random() < 0.75
    alpha = random.random() * 0.2 + 0.9
    random.seed(11)
    feats_vtlp = std(audio)
    std.eval()
    save("g2_frontend_gsc", audio=audio, lengths=lengths, feats=feats, mels=mels,
         lens_in=lens_in, lens_out=std.compute_lengths(lens_in), vtlp_alpha=alpha, mels_vtlp=feats_vtlp[:, 0])
    syn = {}
    for L in (8000, 16000, 13527):
        x = torch.from_numpy(synthetic_clips(L))
        syn[f"audio_{L}"] = x
        syn[f"mels_{L}"] = std(x, mels_only=True)
    save("g2_frontend_synth", **syn)

    # ---- G4: ZMUV ----------------------------------------------------------------------------------
    zmuv = ZmuvTransform()
    for c in clips:
        zmuv.update(std(torch.from_numpy(c)[None]))
    normed = zmuv(feats)
    save("g4_zmuv", total=zmuv.total, mean=zmuv.mean, mean2=zmuv.mean2, std=zmuv.std, normed_ch0=normed[:, 0])

    # ---- G5: res8 ----------------------------------------------------------------------------------
    x = normed
    for C in (4, 12, 30):
        model = RegisteredModel.find_registered_class("res8")(C)
        model.load_state_dict(om.res8_init(C))
        model.eval()
        with torch.no_grad():
            eval_logits = model(x, None)
        labels = torch.arange(x.size(0)) % C
        out = dict(labels=labels, eval_logits=eval_logits)
        model.train()
        params = list(model.parameters())
        opt = torch.optim.AdamW(params, 0.01, weight_decay=1e-5)
        crit = torch.nn.CrossEntropyLoss()
        for step in range(3):
            scores = model(x, None)
            opt.zero_grad(); model.zero_grad()
            loss = crit(scores, labels)
            loss.backward()
            if step == 0:
                out["train_logits"] = scores.detach().clone()
                out["loss0"] = loss.detach().clone()
                for n, p in model.named_parameters():
                    out["grad0." + n] = p.grad.detach().clone()
                for i in (1, 6):
                    out[f"bn{i}.running_mean.1"] = getattr(model, f"bn{i}").running_mean.clone()
                    out[f"bn{i}.running_var.1"] = getattr(model, f"bn{i}").running_var.clone()
            opt.step()
            out[f"loss{step}"] = loss.detach().clone()
        for n, t in model.state_dict().items():
            out["sd3." + n] = t.clone()
        model.eval()
        with torch.no_grad():
            out["eval_logits_after3"] = model(x, None)
        save(f"g5_res8_c{C}", **out)

    # 0.5 s geometry (T=41) eval + one train step, C=4 (BASELINE config 2 shape)
    x05 = zmuv(std(audio[:, 2000:10000]))
    model = RegisteredModel.find_registered_class("res8")(4)
    model.load_state_dict(om.res8_init(4))
    model.eval()
    with torch.no_grad():
        ev = model(x05, None)
    model.train()
    sc = model(x05, None)
    loss = torch.nn.functional.cross_entropy(sc, torch.arange(6) % 4)
    loss.backward()
    save("g5_res8_c4_half", audio=audio[:, 2000:10000], eval_logits=ev, train_logits=sc, loss0=loss,
         **{"grad0.conv0.weight": model.conv0.weight.grad, "grad0.conv1.weight": model.conv1.weight.grad,
            "grad0.conv6.weight": model.conv6.weight.grad, "grad0.output.weight": model.output.weight.grad})

    # ---- G6: lstm / seq-lstm -------------------------------------------------------------------------
    flen = std.compute_lengths(lengths)
    for name in ("lstm", "seq-lstm"):
        model = RegisteredModel.find_registered_class(name)(5)
        model.load_state_dict(om.lstm_init(5))
        model.eval()
        with torch.no_grad():
            logits = model(x, flen)
        out = dict(frame_lengths=flen, logits=logits)
        model.train()
        sc = model(x, flen)
        if name == "lstm":
            loss = torch.nn.functional.cross_entropy(sc, torch.arange(6) % 5)
        else:
            lp = torch.nn.functional.log_softmax(sc, -1)
            targets = torch.tensor([[0, 1, 2]] * 6)
            loss = torch.nn.CTCLoss(4)(lp, targets, flen, torch.tensor([3] * 6))
        loss.backward()
        out["loss0"] = loss.detach()
        for n, p in model.named_parameters():
            out["grad0." + n] = p.grad.detach().clone()
        # streaming carry over two consecutive calls (rnn.py:62,67-68)
        model.eval().streaming()
        with torch.no_grad():
            a = model(x[:1, :, :, :40], None) if name == "seq-lstm" else model(x[:1, :, :, :40], torch.tensor([40]))
            b = model(x[:1, :, :, 40:], None) if name == "seq-lstm" else model(x[:1, :, :, 40:], torch.tensor([41]))
        out["stream_a"], out["stream_b"] = a, b
        save("g6_" + name.replace("-", "_"), **out)

    # ---- G7: SpecAugment with recorded draws -------------------------------------------------------
    class Recorder(random.Random):
        def __init__(self, seed):
            super().__init__(seed)
            self.log = []

        def randrange(self, *a):
            v = super().randrange(*a)
            self.log.append(v)
            return v

    sa = SpecAugmentTransform().train()
    rec = Recorder(5)
    sa.rand = rec
    xin = x.clone()
    # force both augments (prob gates draw from the same RNG via .random())
    for p in sa.augment_params:
        p.prob = 1.1
    for p in sa.augment_params:
        p.current_value_idx = 2
    sa.augment_params[1].domain = [10, 50, 60, 125, 150]  # T=81 > t so the mask is not skipped
    xout = sa(xin)
    B = x.size(0)
    save("g7_specaug", x=x, out=xout, f=np.array(rec.log[0:2 * B:2]), f0=np.array(rec.log[1:2 * B:2]),
         t=np.array(rec.log[2 * B::2]), t0=np.array(rec.log[2 * B + 1::2]))

    # ---- G8: engines -------------------------------------------------------------------------------
    SETTINGS.inference_engine.inference_sequence = [0, 1, 2]
    ctx = InferenceContext(["hey", "fire", "fox"], token_type="word")
    model = RegisteredModel.find_registered_class("res8")(ctx.num_labels)
    model.load_state_dict(om.res8_init(ctx.num_labels))
    model.eval().streaming()
    engine = FrameInferenceEngine(500, 63, model, zmuv, ctx)
    clip = torch.from_numpy(np.concatenate(clips[:3]))
    present = engine.infer(clip)
    hist = np.array(engine.label_history, dtype=np.float64)
    probs = np.stack([p for _, p in engine.pred_history]) if engine.pred_history else np.zeros((0, 4))
    n_windows = sum(1 for _ in stride(clip, 500, 63, 16000))
    save("g8_frame_engine", clip=clip, present=np.array(present), label_history=hist, last_probs=probs,
         n_windows=np.array(n_windows), num_labels=np.array(ctx.num_labels),
         negative_label=np.array(ctx.negative_label), blank_label=np.array(ctx.blank_label))

    ctx_b = InferenceContext(["hey", "fire", "fox"], token_type="word", use_blank=True)
    smodel = RegisteredModel.find_registered_class("seq-lstm")(ctx_b.num_labels)
    smodel.load_state_dict(om.lstm_init(ctx_b.num_labels))
    smodel.eval().streaming()
    SETTINGS.inference_engine.smoothing_window_ms = 0
    eng = InferenceEngine(smodel, zmuv, ctx_b)
    present = eng.infer(torch.from_numpy(clips[0]))
    save("g8_seq_engine", clip=clips[0], present=np.array(present),
         label_history=np.array(eng.label_history, dtype=np.float64), num_labels=np.array(ctx_b.num_labels),
         blank_label=np.array(ctx_b.blank_label))




def collate_protocol_golden():
    """G7b: RNG protocol of the training collate chain (pretrain_gsc.py:78-80): which random.random() draws
    compose(truncate, Timeshift.train(), Noise.train()) consumes, and the crops that result."""
    from functools import partial
    from howl.data.transform.operator import compose, truncate_length
    from howl.data.transform.transform import NoiseTransform, TimeshiftTransform

    class Ex:
        def __init__(self, audio):
            self.audio_data = audio

        def update_audio_data(self, audio, **kw):
            return Ex(audio)

    draws = []
    real = random.random

    def recording():
        v = real()
        draws.append(v)
        return v

    lens = [16000, 12971, 16000, 14336, 9000, 20000]
    out = {}
    for trial, seed in enumerate((0, 1, 2, 5)):
        random.seed(seed)
        torch.manual_seed(seed)
        draws.clear()
        random.random = recording
        try:
            chain = compose(partial(truncate_length, length=16000), TimeshiftTransform().train(), NoiseTransform().train())
            # ramp signals make the crop offset observable
            exs = chain([Ex(torch.arange(L, dtype=torch.float32) * 1e-5) for L in lens])
        finally:
            random.random = real
        out[f"draws_{trial}"] = np.array(draws)
        out[f"out_len_{trial}"] = np.array([e.audio_data.numel() for e in exs])
        out[f"first_{trial}"] = np.array([float(e.audio_data[0]) for e in exs])   # ~ offset * 1e-5 (+ noise <= ~0.004)
    save("g7b_collate_protocol", lens=np.array(lens), **out)


def mixer_golden():
    """G9: DatasetMixer (transform.py:199-231) in front of the training collate chain, as train.py:218 composes it:
    outputs of the reference class itself for a seeded ``random`` stream, ramp waveforms and constant-level backgrounds
    (so the chosen clip, its offset and alpha are all readable from the result)."""
    from howl.data.transform.transform import DatasetMixer

    class Ex:
        def __init__(self, audio):
            self.audio_data = audio

        def update_audio_data(self, audio, **kw):
            return Ex(audio)

    wf_lens = [16000, 12971, 16000, 9000]
    bg_lens = [40000, 8000, 25000, 16000, 31000]
    out = {}
    for trial, seed in enumerate((0, 3, 11)):
        random.seed(seed)
        bgs = [Ex(torch.full((L,), 0.1 * (i + 1)) + torch.arange(L, dtype=torch.float32) * 1e-6) for i, L in enumerate(bg_lens)]
        mixer = DatasetMixer(bgs).train()
        exs = mixer([Ex(torch.arange(L, dtype=torch.float32) * 1e-5) for L in wf_lens])
        for i, e in enumerate(exs):
            a = e.audio_data.numpy().astype(np.float32)
            out[f"mixed_{trial}_{i}"] = np.concatenate([a[:8], a[8:-8:61], a[-8:]])   # subsampled: head, every 61st, tail
        out[f"next_draw_{trial}"] = np.array(random.random())      # position of the random stream afterwards
    save("g9_mixer", wf_lens=np.array(wf_lens), bg_lens=np.array(bg_lens), **out)


FRAME_CLIP_LENS = [30000, 5000, 40000, 16000, 9000, 24000, 32000, 8000, 12000, 20000]
FRAME_LABEL_MAPS = [{450.0: 0, 900.0: 1, 1400.0: 2}, {}, {}, {300.0: 1}, {120.5: 0, 480.0: 2}, {1450.0: 2, 700.0: 0},
                    {1999.0: 1}, {}, {10.0: 0}, {600.0: 3, 1100.0: 0}]


def frame_clip(i, L):
    """Ramp with a per-clip base: a sample's value names (clip, offset)."""
    return torch.arange(L, dtype=torch.float32) * 1e-6 + 0.05 * (i + 1)


def frame_batchifier_golden():
    """G10: WakeWordFrameBatchifier (batchifier.py:37-118) + tensorize_audio_data(rand_append) (operator.py:89-109), outputs
    of the reference class itself under a seeded ``random`` stream; ramp clips make clip / offset / padding side readable."""
    from types import SimpleNamespace
    from howl.data.transform.batchifier import WakeWordFrameBatchifier

    class Ex:
        def __init__(self, audio, tl):
            self.audio_data, self.label_data = audio, SimpleNamespace(timestamp_label_map=tl)

        def update_audio_data(self, audio, **kw):
            return Ex(audio, self.label_data)

    out = {"clip_lens": np.array(FRAME_CLIP_LENS),          # the label maps in CSR form (insertion order kept)
           "map_ptr": np.cumsum([0] + [len(m) for m in FRAME_LABEL_MAPS]),
           "map_end_ms": np.array([k for m in FRAME_LABEL_MAPS for k in m], np.float64),
           "map_label": np.array([v for m in FRAME_LABEL_MAPS for v in m.values()], np.int64)}
    variants = [dict(), dict(), dict(positive_sample_prob=0.8, window_size_ms=1000), dict(pad_to_window=False, window_size_ms=250)]
    for trial, (seed, kw) in enumerate(zip((0, 7, 21, 3), variants)):
        random.seed(seed)
        fb = WakeWordFrameBatchifier(4, **kw)
        batch = fb([Ex(frame_clip(i, L), dict(tl)) for i, (L, tl) in enumerate(zip(FRAME_CLIP_LENS, FRAME_LABEL_MAPS))])
        a = batch.audio_data.numpy()
        nz = a != 0
        out[f"labels_{trial}"] = batch.labels.numpy()
        out[f"lengths_{trial}"] = batch.lengths.numpy()
        out[f"width_{trial}"] = np.array(a.shape[1])
        out[f"nz_start_{trial}"] = np.array([int(r.argmax()) if r.any() else -1 for r in nz])
        out[f"nz_count_{trial}"] = nz.sum(1)
        out[f"first_{trial}"] = np.array([r[m][0] if m.any() else 0.0 for r, m in zip(a, nz)], np.float32)
        out[f"last_{trial}"] = np.array([r[m][-1] if m.any() else 0.0 for r, m in zip(a, nz)], np.float32)
        out[f"every53_{trial}"] = a[:, ::53]
        out[f"next_draw_{trial}"] = np.array(random.random())
    save("g10_frame_batchifier", **out)


PHONE_DICT_TEXT = """;;; test pronunciation dictionary (CMUdict layout)
HEY  HH EY1
HEY(2)  HH EH1
FIRE  F AY1 ER0
FOX  F AA1 K S
FIREFOX  F AY1 ER0 F AA1 K S
HELLO  HH AH0 L OW1
WORLD  W ER1 L D
IT'S  IH1 T S
A  AH0
PAUSE  sil
"""
PHONE_VOCAB = ["hey", "fire", "fox"]
PHONE_TRANSCRIPTS = ["hey fire fox", "hello world hey firefox", "it\u2019s a fox, hey! pause fire fox", "<unk> hey hey fire",
                     "world helloworld xyzzy fox"]
PHONE_QUERIES = ["hh ey1 f ay1 er0 f aa1 k s", "hh ey1 sil f ay1 er0 sp f aa1 k s", "f aa1 k s hh ey1", "hh ah0 l ow1",
                 "f ay1 er0 f aa1 k s", "hh ey1 f ay1 er0 w er1 l d f aa1 k s", ""]


def phone_context_golden():
    """G11: InferenceContext(token_type="phone") (context.py:52-99) with its PhoneticFrameLabeler / PhoneticTranscriptSearcher /
    LabelColoring, and the PhonePhrase index helpers -- outputs of the reference classes on a small dictionary."""
    import json
    import tempfile
    from types import SimpleNamespace
    from howl.context import InferenceContext
    from howl.data.common.phone import PhonePhrase
    from howl.settings import SETTINGS
    with tempfile.NamedTemporaryFile("w", suffix=".dict", delete=False) as f:
        f.write(PHONE_DICT_TEXT)
    SETTINGS.training.phone_dictionary = f.name
    SETTINGS.inference_engine.inference_sequence = [0, 1, 2]
    out = {}
    for tag, use_blank in (("noblank", False), ("blank", True)):
        ctx = InferenceContext(PHONE_VOCAB, token_type="phone", use_blank=use_blank)
        out[tag] = {"adjusted_vocab": list(ctx.adjusted_vocab), "num_labels": ctx.num_labels, "negative_label": ctx.negative_label,
                    "blank_label": ctx.blank_label, "color_map": {str(k): v for k, v in ctx.coloring.color_map.items()},
                    "pattern": ctx.searcher.pattern.pattern}
    labels = []
    for tr in PHONE_TRANSCRIPTS:
        md = SimpleNamespace(transcription=tr, end_timestamps=[10.0 * (i + 1) for i in range(len(tr))])
        labels.append({str(k): v for k, v in ctx.labeler.compute_frame_labels(md).timestamp_label_map.items()})
    out["frame_labels"] = labels
    out["search"] = [bool(ctx.searcher.search(q)) for q in PHONE_QUERIES]
    out["contains_any"] = [bool(ctx.searcher.contains_any(q)) for q in PHONE_QUERIES]
    pp = PhonePhrase.from_string("hh ey1 sil f ay1 sp er0 spn")
    out["phrase"] = {"audible": pp.audible_transcript, "sil": pp.sil_indices,
                     "all_to_transcript": [pp.all_idx_to_transcript_idx(i) for i in range(len(pp.phones))],
                     "audible_to_all": [pp.audible_idx_to_all_idx(i) for i in range(len(pp.audible_phones))],
                     "index_er0": pp.audible_index(PhonePhrase.from_string("sil er0")),
                     "index_from1": pp.audible_index(PhonePhrase.from_string("ay1 er0"), 1)}
    out["inputs"] = {"dictionary": PHONE_DICT_TEXT, "vocab": PHONE_VOCAB, "transcripts": PHONE_TRANSCRIPTS, "queries": PHONE_QUERIES}
    words = {}
    for w in ("firefox", "helloworld", "heyfox", "it's"):
        words[w] = str(ctx.labeler.transform(w))
    out["transform"] = words
    (HERE / "g11_phone_context.json").write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")
    print("g11_phone_context.json", {k: (len(v) if hasattr(v, "__len__") else v) for k, v in out.items()})


def checkpoint_golden():
    """G12 (f4): a workspace WRITTEN BY THE REFERENCE -- ``howl.workspace.Workspace.save_model`` (workspace.py:56-63:
    ``torch.save(model.state_dict())``) for a res8 after two AdamW steps (non-trivial weights and BatchNorm buffers), and
    ``zmuv.pt.bin`` as ``pretrain_gsc.py:106`` writes it -- copied to ``tests/golden/ref_workspace/`` as data, together with
    the eval logits the reference computes after loading that workspace back (``Workspace.load_model``, hubconf.py:53-84)."""
    import shutil
    import tempfile
    import types
    _reference_shims._module("torch.utils.tensorboard",
                             SummaryWriter=lambda *a, **k: types.SimpleNamespace(add_scalar=lambda *a, **k: None))
    from howl.workspace import Workspace
    C = 12
    clips = [read_wav(GSC / w) for w in WAVS]
    audio, _ = batchify_like_reference(clips)
    std = StandardAudioTransform().eval()
    zmuv = ZmuvTransform()
    for c in clips:
        zmuv.update(std(torch.from_numpy(c)[None]))
    x = zmuv(std(audio))
    model = RegisteredModel.find_registered_class("res8")(C)
    model.load_state_dict(om.res8_init(C))
    model.train()
    opt = torch.optim.AdamW(model.parameters(), 0.01, weight_decay=1e-5)
    labels = torch.arange(x.size(0)) % C
    for _ in range(2):
        opt.zero_grad()
        torch.nn.functional.cross_entropy(model(x, None), labels).backward()
        opt.step()
    with tempfile.TemporaryDirectory() as tmp:
        ws = Workspace(Path(tmp) / "ws", delete_existing=False)
        ws.save_model(model, best=True)
        torch.save(zmuv.state_dict(), str(ws.path / "zmuv.pt.bin"))
        fresh = RegisteredModel.find_registered_class("res8")(C)
        ws.load_model(fresh, best=True)
        fresh.eval()
        z2 = ZmuvTransform()
        z2.load_state_dict(torch.load(str(ws.path / "zmuv.pt.bin")))
        with torch.no_grad():
            logits = fresh(z2(std(audio)), None)
        dst = HERE / "ref_workspace"
        dst.mkdir(exist_ok=True)
        for name in ("model-best.pt.bin", "zmuv.pt.bin"):
            shutil.copyfile(ws.path / name, dst / name)
    sd = fresh.state_dict()
    save("g12_ref_workspace", audio=audio, eval_logits=logits, num_labels=np.array(C),
         keys=np.array(list(sd.keys())), shapes=np.array([str(tuple(v.shape)) for v in sd.values()]),
         dtypes=np.array([str(v.dtype) for v in sd.values()]),
         zmuv_keys=np.array(list(z2.state_dict().keys())))


def wide_golden():
    """G13: the reference at its STOCK mel count (``NUM_MELS`` unset -> 80, settings.py:32) -- filterbanks, the frontend's
    output for the six GSC clips (eval and one VTLP draw), ZMUV statistics, and res8 (12 labels) on those features: eval logits,
    one training step's logits / loss / gradients / BatchNorm buffers, the logits after the AdamW step."""
    old = SETTINGS.audio_transform.num_mels
    SETTINGS.audio_transform.num_mels = 80
    try:
        clips = [read_wav(GSC / w) for w in WAVS]
        audio, _ = batchify_like_reference(clips)
        std = StandardAudioTransform().eval()
        out = {"audio": audio,
               "fb_standard": create_vtlp_fb_matrix(257, 0.0, 8000.0, 80, 16000, 1.0, training=False),
               "fb_vtlp_1.0999": create_vtlp_fb_matrix(257, 0.0, 8000.0, 80, 16000, 1.0999, training=True),
               "fb_vtlp_0.9": create_vtlp_fb_matrix(257, 0.0, 8000.0, 80, 16000, 0.9, training=True)}
        feats = std(audio)
        assert feats.shape == (6, 3, 80, 81)
        out["feats"] = feats
        std.train()
        random.seed(11)
        assert random.random() < 0.75
        out["vtlp_alpha"] = random.random() * 0.2 + 0.9
        random.seed(11)
        out["mels_vtlp"] = std(audio)[:, 0]
        std.eval()
        zmuv = ZmuvTransform()
        for c in clips:
            zmuv.update(std(torch.from_numpy(c)[None]))
        out["zmuv_mean"], out["zmuv_mean2"] = zmuv.mean, zmuv.mean2
        x = zmuv(feats)
        C = 12
        model = RegisteredModel.find_registered_class("res8")(C)
        model.load_state_dict(om.res8_init(C))
        model.eval()
        with torch.no_grad():
            out["eval_logits"] = model(x, None)
        labels = torch.arange(x.size(0)) % C
        model.train()
        opt = torch.optim.AdamW(model.parameters(), 0.01, weight_decay=1e-5)
        scores = model(x, None)
        loss = torch.nn.CrossEntropyLoss()(scores, labels)
        loss.backward()
        out["train_logits"], out["loss0"], out["labels"] = scores.detach().clone(), loss.detach().clone(), labels
        for n, p in model.named_parameters():
            out["grad0." + n] = p.grad.detach().clone()
        for i in (1, 6):
            out[f"bn{i}.running_mean.1"] = getattr(model, f"bn{i}").running_mean.clone()
            out[f"bn{i}.running_var.1"] = getattr(model, f"bn{i}").running_var.clone()
        opt.step()
        model.eval()
        with torch.no_grad():
            out["eval_logits_after1"] = model(x, None)
        save("g13_res8_80mel", **out)
    finally:
        SETTINGS.audio_transform.num_mels = old


def long_window_golden():
    """G14: the reference's res8 on 2-s windows (MAX_WINDOW_SIZE_SECONDS=2: 161 frames, cnn.py:127-145 takes any T) -- pairs of the
    GSC clips concatenated, 40 and 80 mel bins: eval logits, one training step's logits / loss / gradients / BatchNorm buffers."""
    clips = [read_wav(GSC / w) for w in WAVS]
    pairs = [np.concatenate([np.pad(clips[i], (0, 16000 - len(clips[i]))), np.pad(clips[j], (0, 16000 - len(clips[j])))])
             for i, j in ((0, 3), (4, 1), (2, 5))]
    audio = torch.from_numpy(np.stack(pairs))
    out = {"audio": audio}
    old = SETTINGS.audio_transform.num_mels
    try:
        for mels in (40, 80):
            SETTINGS.audio_transform.num_mels = mels
            std = StandardAudioTransform().eval()
            zmuv = ZmuvTransform()
            for c in clips:
                zmuv.update(std(torch.from_numpy(c)[None]))
            x = zmuv(std(audio))
            assert x.shape == (3, 3, mels, 161)
            C = 12
            model = RegisteredModel.find_registered_class("res8")(C)
            model.load_state_dict(om.res8_init(C))
            model.eval()
            pre = f"m{mels}."
            out[pre + "x"] = x[:, :1].clone()
            with torch.no_grad():
                out[pre + "eval_logits"] = model(x, None)
            labels = torch.arange(x.size(0)) % C
            model.train()
            scores = model(x, None)
            loss = torch.nn.CrossEntropyLoss()(scores, labels)
            loss.backward()
            out[pre + "train_logits"], out[pre + "loss0"] = scores.detach().clone(), loss.detach().clone()
            for n, p in model.named_parameters():
                out[pre + "grad0." + n] = p.grad.detach().clone()
            for i in (1, 6):
                out[pre + f"bn{i}.running_mean.1"] = getattr(model, f"bn{i}").running_mean.clone()
                out[pre + f"bn{i}.running_var.1"] = getattr(model, f"bn{i}").running_var.clone()
    finally:
        SETTINGS.audio_transform.num_mels = old
    save("g14_res8_two_second_windows", **out)


def whole_clip_sequence_golden():
    """G15: the recurrent models on WHOLE CLIPS, as the reference's sequence objective feeds them -- AudioSequenceBatchifier
    (batchifier.py:14-34) batches clips, not windows, and CTCLoss runs over all their frames (train.py:198-200, 291-296).
    Four clips of 4.0 / 3.25 / 2.6 / 1.6 s made of the GSC clips back to back (318 / 258 / 206 / 128 valid frames of 12.5 ms: three,
    three, two and exactly one 128-frame window of the CTC kernel; 8 x the 41 frames of G6): eval logits, one training step's loss and gradients (seq-lstm: CTC with
    targets of 3 / 2 / 1 / 3 labels, one of them with a repeated label; lstm: cross-entropy), and the streaming carry over a
    160 + 161-frame split of the longest clip (rnn.py:62,67-68)."""
    raw = [read_wav(GSC / w) for w in WAVS]
    clips = [np.pad(c, (0, 16000))[:16000] for c in raw]
    cat = [np.concatenate([clips[0], clips[3], clips[1], clips[4]]), np.concatenate([clips[4], clips[2], clips[5], clips[0]])[:52000],
           np.concatenate([clips[5], clips[0], clips[4]])[:41600], np.concatenate([clips[2], clips[3]])[:26000]]
    audio, lengths = batchify_like_reference(cat)
    std = StandardAudioTransform().eval()
    zmuv = ZmuvTransform()
    for c in raw:                   # (G4's statistics)
        zmuv.update(std(torch.from_numpy(c)[None]))
    x = zmuv(std(audio))
    flen = std.compute_lengths(lengths)
    assert x.shape == (4, 3, 40, 321) and flen.tolist() == [318, 258, 206, 128]   # (the last: exactly one CTC window)
    targets = torch.tensor([[0, 1, 2], [3, 3, 0], [2, 0, 0], [1, 0, 3]])
    tlen = torch.tensor([3, 2, 1, 3])
    for name in ("lstm", "seq-lstm"):
        model = RegisteredModel.find_registered_class(name)(5)
        model.load_state_dict(om.lstm_init(5))
        model.eval()
        with torch.no_grad():
            logits = model(x, flen)
        out = dict(lengths=lengths, x=x[:, :1].clone(), frame_lengths=flen, logits=logits, targets=targets, target_lengths=tlen)
        if name == "seq-lstm":
            out["audio"] = audio            # (once: the same clips for both models)
        model.train()
        sc = model(x, flen)
        if name == "lstm":
            loss = torch.nn.functional.cross_entropy(sc, torch.arange(4) % 5)
        else:
            loss = torch.nn.CTCLoss(4)(torch.nn.functional.log_softmax(sc, -1), targets, flen, tlen)
        loss.backward()
        out["loss0"] = loss.detach()
        for n, p in model.named_parameters():
            out["grad0." + n] = p.grad.detach().clone()
        model.eval().streaming()
        with torch.no_grad():
            a = model(x[:1, :, :, :160], None) if name == "seq-lstm" else model(x[:1, :, :, :160], torch.tensor([160]))
            b = model(x[:1, :, :, 160:], None) if name == "seq-lstm" else model(x[:1, :, :, 160:], torch.tensor([161]))
        out["stream_a"], out["stream_b"] = a, b
        save("g15_whole_clips_" + name.replace("-", "_"), **out)


if __name__ == "__main__":
    only = [a for a in sys.argv[1:] if a.startswith("--only-")]
    if not only:
        main()
    if not only or "--only-collate" in only:
        collate_protocol_golden()
    if not only or "--only-mixer" in only:
        mixer_golden()
    if not only or "--only-frame-batchifier" in only:
        frame_batchifier_golden()
    if not only or "--only-phone" in only:
        phone_context_golden()
    if not only or "--only-checkpoint" in only:
        checkpoint_golden()
    if not only or "--only-wide" in only:
        wide_golden()
    if not only or "--only-long" in only:
        long_window_golden()
    if not only or "--only-clips" in only:
        whole_clip_sequence_golden()
