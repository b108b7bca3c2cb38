"""-m gpu: inference engines and the pretrain_gsc entry point on the HIP path."""
import numpy as np
import pytest
import torch

from gpu_util import DEV, make_res8, t

pytestmark = pytest.mark.gpu


def test_frame_engine_matches_reference_history(golden):
    """G8: label history of FrameInferenceEngine.infer on a fixture clip (500 ms window / 63 ms stride), captured from
    the reference's engine with the same closed-form res8 weights; the batched engine must reproduce it exactly."""
    from howl_amd.context import InferenceContext
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.model.inference import FrameInferenceEngine
    from howl_amd.settings import SETTINGS
    g, g4 = golden("g8_frame_engine"), golden("g4_zmuv")
    SETTINGS.inference_engine.inference_sequence = [0, 1, 2]
    ctx = InferenceContext(["hey", "fire", "fox"], token_type="word")
    assert (ctx.num_labels, ctx.negative_label, ctx.blank_label) == (int(g["num_labels"]), int(g["negative_label"]),
                                                                     int(g["blank_label"]))
    model = make_res8(ctx.num_labels, train=False).streaming()
    zmuv = ZmuvTransform().to(DEV)
    zmuv.mean.copy_(t(g4["mean"]))
    zmuv.mean2.copy_(t(g4["mean2"]))
    zmuv.total.copy_(t(g4["total"]))
    engine = FrameInferenceEngine(500, 63, model, zmuv, ctx)
    clip = t(g["clip"]).to(DEV)
    probs = engine.window_probabilities(clip)
    assert probs.shape[0] == int(g["n_windows"])
    present = engine.infer(clip)
    assert bool(present) == bool(g["present"])
    hist = np.array(engine.label_history, dtype=np.float64)
    assert hist.shape == g["label_history"].shape
    assert np.array_equal(hist, g["label_history"])          # timestamps and label indices, bit-exact
    assert np.abs(engine.pred_history[-1][1] - g["last_probs"][-1]).max() < 1e-4
    # sequential path (what the live client calls per frame) gives the same labels
    engine.reset()
    labels = [engine.ingest_frame(clip[i * 1008: i * 1008 + 8000], curr_time=63.0 * i) for i in range(5)]
    assert labels == [int(x) for x in g["label_history"][:5, 1]]
    # several clips at once (an evaluation pass: one frontend launch, one forward, one host copy for all of their windows):
    # the same probabilities per clip -- ragged lengths, a clip shorter than a window, one too short for any -- and the same verdicts
    others = [clip[3000:40000], clip[:6000], clip[:500], clip]
    many = engine.window_probabilities_many(others)
    for c, p in zip(others, many):
        engine.reset()
        single = engine.window_probabilities(c)
        assert p.shape == single.shape and (p.size == 0 or np.abs(p - single).max() < 1e-6)
        assert np.array_equal(p.argmax(1), single.argmax(1))
    singles = []
    for c in others:
        engine.reset()
        singles.append(bool(engine.infer(c)))
    assert engine.infer_many(others) == singles and singles[-1] == bool(g["present"])
    # a cap on the windows per launch (an evaluation pass must stay O(cap) in memory): same numbers in pieces
    engine.MAX_WINDOWS_PER_LAUNCH = 7
    capped = engine.window_probabilities_many(others)
    for p, q in zip(many, capped):
        assert p.shape == q.shape and (p.size == 0 or np.abs(p - q).max() < 1e-6)
    assert engine.infer_many(others) == singles


def test_sequence_engine_matches_reference_history(golden):
    """G8: InferenceEngine.infer (seq-lstm with a CTC blank, whole clip in one forward, smoothing window 0) -- the label history
    the reference's engine produced with the same closed-form weights must be reproduced exactly."""
    from howl_amd.context import InferenceContext
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.model import RegisteredModel
    from howl_amd.model.inference import InferenceEngine
    from howl_amd.settings import SETTINGS
    from oracle import models as om
    g, g4 = golden("g8_seq_engine"), golden("g4_zmuv")
    SETTINGS.inference_engine.inference_sequence = [0, 1, 2]
    SETTINGS.inference_engine.smoothing_window_ms = 0
    try:
        ctx = InferenceContext(["hey", "fire", "fox"], token_type="word", use_blank=True)
        assert (ctx.num_labels, ctx.blank_label) == (int(g["num_labels"]), int(g["blank_label"]))
        model = RegisteredModel.find_registered_class("seq-lstm")(ctx.num_labels)
        model.load_state_dict(om.lstm_init(ctx.num_labels))
        model = model.to(DEV).eval().streaming()
        zmuv = ZmuvTransform().to(DEV)
        zmuv.mean.copy_(t(g4["mean"]))
        zmuv.mean2.copy_(t(g4["mean2"]))
        zmuv.total.copy_(t(g4["total"]))
        engine = InferenceEngine(model, zmuv, ctx)
        present = engine.infer(t(g["clip"]).to(DEV))
        assert bool(present) == bool(g["present"])
        hist = np.array(engine.label_history, dtype=np.float64)
        assert hist.shape == g["label_history"].shape
        assert np.array_equal(hist[:, 1], g["label_history"][:, 1])              # label indices, bit-exact
        assert np.abs(hist[:, 0] - g["label_history"][:, 0]).max() < 1e-6        # frame times
    finally:
        SETTINGS.reset()


@pytest.mark.parametrize("num_mels", ["40", None])
def test_pretrain_gsc_entry_point_synthetic(tmp_path, monkeypatch, num_mels):
    """`python -m training.run.pretrain_gsc --model res8` flow on generated clips: ZMUV pass, fused training epochs,
    dev accuracy, workspace artefacts with the reference's file names and state_dict keys.  NUM_MELS=40 as envs/res8.env sets
    it, and UNSET: the reference's stock default of 80 bins (settings.py:32), what a user without the preset file runs."""
    for k, v in dict(NUM_EPOCHS="4", BATCH_SIZE="64", MAX_WINDOW_SIZE_SECONDS="1", LEARNING_RATE="0.01", LR_DECAY="0.8",
                     DEVICE="cuda:0").items():
        monkeypatch.setenv(k, v)
    if num_mels is None:
        monkeypatch.delenv("NUM_MELS", raising=False)
    else:
        monkeypatch.setenv("NUM_MELS", num_mels)
    from howl_amd.settings import SETTINGS
    SETTINGS.reset()
    assert SETTINGS.audio_transform.num_mels == (80 if num_mels is None else 40)
    try:
        from howl_amd.training.run import pretrain_gsc
        ws = tmp_path / "ws"
        pretrain_gsc.main(["--model", "res8", "--workspace", str(ws), "--synthetic", "512"])
        for name in ("model.pt.bin", "model-best.pt.bin", "zmuv.pt.bin", "settings.json", "cmd-args.json"):
            assert (ws / name).exists(), name
        sd = torch.load(ws / "model-best.pt.bin")
        assert list(sd)[:2] == ["conv0.weight", "bn1.running_mean"] and sd["output.weight"].shape == (30, 45)
        assert set(torch.load(ws / "zmuv.pt.bin")) == {"total", "mean", "mean2"}
        import json
        lines = [json.loads(l) for l in (ws / "logs" / "scalars.jsonl").read_text().splitlines()]
        losses = [l["value"] for l in lines if l["tag"] == "Training/Loss"]
        assert len(losses) == 32 and min(losses[-8:]) < losses[0]
        accs = [l["value"] for l in lines if l["tag"] == "Dev/Metric/acc"]
        assert len(accs) == 4 and max(accs) > 1.0 / 30     # above chance after a handful of steps on separable tones
    finally:
        monkeypatch.setenv("NUM_MELS", "40")      # (tests/conftest.py's default for everything else)
        SETTINGS.reset()


def test_device_collate_batch():
    """a12: truncate -> timeshift -> noise -> batchify on the device: sorted by length, zero padded, clamped, labelled."""
    import random
    from howl_amd.data.collate import DeviceCollate
    from howl_amd.utils.synth import synthetic_pcm
    pcm = synthetic_pcm(32, 16000).to(DEV)
    lengths = torch.tensor([16000 - 311 * (i % 7) for i in range(32)])
    labels = (torch.arange(32) % 5).to(DEV)
    random.seed(1)
    dc = DeviceCollate(pcm, lengths, labels, 16000, seed=None)
    batch = dc(list(range(32)))
    L = batch.lengths.cpu()
    assert (L[:-1] >= L[1:]).all() and batch.audio_data.shape == (32, int(L.max()))
    assert batch.audio_data.abs().max().item() <= 1.0
    for row in (0, 31):
        assert not batch.audio_data[row, int(L[row]):].any()
    assert batch.labels.shape == (32,)
    dc.training = False        # eval: no augmentation, plain batchify
    plain = dc(list(range(32)))
    order = sorted(range(32), key=lambda k: -int(lengths[k]))
    assert torch.equal(plain.audio_data[0, : int(lengths[order[0]])], pcm[order[0], : int(lengths[order[0]])])


@pytest.mark.parametrize("model,objective,mels,window", [("res8", "frame", "40", "0.5"), ("seq-lstm", "ctc", "40", "0.5"),
                                                         ("mobilenet", "frame", "40", "0.5"), ("res8", "frame", "80", "1.5")])
def test_train_entry_point_synthetic(tmp_path, monkeypatch, model, objective, mels, window):
    """`python -m training.run.train` flow (envs/res8.env / envs/seq-lstm.env presets, shortened) on generated wake-word
    clips: runs end to end, loss goes down, detection results + workspace artefacts are written.  Last case: res8 at the stock 80 mel
    bins with 1.5-s training windows (121 frames: two column x two row strips per utterance in the kernels)."""
    env = dict(NUM_EPOCHS="3", BATCH_SIZE="16", MAX_WINDOW_SIZE_SECONDS=window, LEARNING_RATE={"res8": "0.01", "mobilenet": "0.001"}.get(model, "0.002"),
               LR_DECAY="0.955", WEIGHT_DECAY="0.00001", NUM_MELS=mels, DEVICE="cuda:0", OBJECTIVE=objective,
               TOKEN_TYPE="word", VOCAB='["hey","fire","fox"]', INFERENCE_SEQUENCE="[0,1,2]", INFERENCE_THRESHOLD="0",
               SMOOTHING_WINDOW_MS="0" if objective == "ctc" else "50")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    from howl_amd.settings import SETTINGS
    SETTINGS.reset()
    from howl_amd.training.run import train
    ws = tmp_path / "ws"
    # --eval-freq 1: an in-training evaluation after every epoch (streaming state left behind by the engines must not leak
    # into the next epoch's batches; train.py:284)
    pos, neg = train.main(["--model", model, "--workspace", str(ws), "--synthetic", "96", "--eval-freq", "1"])
    assert pos["tp"] + pos["fn"] == 32 and neg["fp"] + neg["tn"] == 32
    import json
    lines = [json.loads(l) for l in (ws / "logs" / "scalars.jsonl").read_text().splitlines()]
    losses = [l["value"] for l in lines if l["tag"] == "Training/Loss"]
    assert len(losses) == 3 and losses[-1] < losses[0]
    assert (ws / "model.pt.bin").exists() and (ws / "zmuv.pt.bin").exists() and (ws / "0.0_results.csv").exists()
    monkeypatch.setenv("NUM_MELS", "40")
    SETTINGS.reset()


@pytest.mark.parametrize("model,objective", [("res8", "frame"), ("seq-lstm", "ctc")])
def test_train_entry_point_on_a_howl_format_dataset(tmp_path, monkeypatch, model, objective):
    """`python -m training.run.train -i DS`: a dataset directory in the reference's layout (aligned-metadata-*.jsonl +
    audio/*.wav, dataset_loader.py:34-70) is decoded into the device clip bank, labelled by the context's frame labeler and
    trained on through the device collate chain.  OBJECTIVE=ctc: the clips are 2 - 4 s long and go through the sequence
    batchifier WHOLE (batchifier.py:14-34, train.py:198-200): 160 - 320 frames per utterance through the recurrences and the
    CTC kernel's 128-frame windows, no vendor loss kernel anywhere (torch's ctc_loss is made to raise for the run)."""
    import json
    import wave
    import numpy as np
    from types import SimpleNamespace
    env = dict(NUM_EPOCHS="2", BATCH_SIZE="16", MAX_WINDOW_SIZE_SECONDS="0.5", LEARNING_RATE="0.01", LR_DECAY="0.955",
               WEIGHT_DECAY="0.00001", NUM_MELS="40", DEVICE="cuda:0", OBJECTIVE=objective, TOKEN_TYPE="word",
               VOCAB='["hey","fire","fox"]', INFERENCE_SEQUENCE="[0,1,2]", INFERENCE_THRESHOLD="0",
               SMOOTHING_WINDOW_MS="0" if objective == "ctc" else "50")
    if objective == "ctc":
        env["LEARNING_RATE"] = "0.002"
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    from howl_amd.settings import SETTINGS
    SETTINGS.reset()
    from howl_amd.training.run import train

    def no_vendor_ctc(*a, **k):
        raise AssertionError("torch.nn.functional.ctc_loss on the training path")
    monkeypatch.setattr(torch.nn.functional, "ctc_loss", no_vendor_ctc)
    ds = tmp_path / "ds"
    (ds / "audio").mkdir(parents=True)
    rng = np.random.default_rng(0)
    vocab = ["hey", "fire", "fox"]
    for split, n in (("training", 48), ("dev", 12), ("test", 4)):
        with (ds / f"aligned-metadata-{split}.jsonl").open("w") as f:
            for i in range(n):
                ids = [0, 1, 2] if i % 2 == 0 else [int(v) for v in rng.permutation(3)[: 1 + i % 3]]
                if ids == [0, 1, 2] and i % 2:
                    ids = [2, 1, 0]
                pcm, meta = train.make_clip(ids, vocab, rng)
                if objective == "ctc":      # whole clips of 2 - 4 s: a quiet tail behind the words
                    tail = int(rng.integers(int(2.0 * 16000), int(4.0 * 16000))) - pcm.numel()
                    pcm = torch.cat([pcm, 0.01 * torch.from_numpy(rng.standard_normal(max(tail, 0)).astype(np.float32))])
                name = f"{split}_{i}.wav"
                with wave.open(str(ds / "audio" / name), "wb") as w:
                    w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
                    w.writeframes((pcm.numpy().clip(-1, 1) * 32767).astype("<i2").tobytes())
                f.write(json.dumps(dict(path=name, transcription=meta.transcription, end_timestamps=meta.end_timestamps,
                                        phone_strings=None, words=None, phone_end_timestamps=None)) + "\n")
    ws = tmp_path / "ws"
    pos, neg = train.main(["--model", model, "--workspace", str(ws), "-i", str(ds), "--eval-freq", "1"])
    assert pos["tp"] + pos["fn"] == 6 and neg["fp"] + neg["tn"] == 6
    lines = [json.loads(l) for l in (ws / "logs" / "scalars.jsonl").read_text().splitlines()]
    losses = [l["value"] for l in lines if l["tag"] == "Training/Loss"]
    assert len(losses) == 2 and all(v == v for v in losses)
    assert (ws / "model.pt.bin").exists() and (ws / "zmuv.pt.bin").exists()
    SETTINGS.reset()
