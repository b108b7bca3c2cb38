"""``python -m training.run.train --model res8 --workspace W [--load-weights --load-last --eval-freq N --eval]`` on MI355X.

The training / evaluation flow of ``training/run/train.py:35-324`` of the reference for the two objectives its presets
use (``envs/res8.env``: frame-level cross-entropy; ``envs/seq-lstm.env``: CTC): InferenceContext label space, ZMUV pass,
``model.streaming()``, per-step  frontend -> ZMUV -> SpecAugment -> model -> loss -> AdamW, per-epoch LR decay, evaluation
every ``--eval-freq`` epochs through ``FrameInferenceEngine`` / ``InferenceEngine`` with TP/FN (positive clips) and FP/TN
(negative clips) counts written to ``<threshold>_results.csv`` like ``train.py:66-94``.

Datasets: the reference's aligned-metadata dataset stack (``howl/data/dataset``, ``howl/dataset*``) is disk I/O outside the
MI355X hot path; this entry point trains on generated wake-word clips (``--synthetic N``): word w of the vocabulary is a
tone burst of its own pitch, positives are the full sequence, negatives are shuffled / partial sequences, and the frame
batchifier semantics (window ending at a word's end timestamp -> that word's label, else the negative label;
``batchifier.py:56-118``) are applied to the known burst boundaries.
"""
import argparse
import csv
import random
from pathlib import Path

import numpy as np
import torch

from howl_amd import ops
from howl_amd.context import InferenceContext
from howl_amd.data.transform.operator import ZmuvTransform
from howl_amd.data.transform.transform import SpecAugmentTransform, StandardAudioTransform
from howl_amd.model import RegisteredModel
from howl_amd.model.inference import FrameInferenceEngine, InferenceEngine
from howl_amd.settings import SETTINGS
from howl_amd.training.fused import FusedRes8Trainer
from howl_amd.utils.random_utils import set_random_seed
from howl_amd.workspace import Workspace

SR = 16000
WORD_S = 0.3     # seconds per word burst
GAP_S = 0.08


def make_clip(words, n_vocab, rng):
    """Tone bursts for the word ids in ``words``; returns (pcm, [(word, end_sample), ...])."""
    parts, ends, pos = [np.zeros(int(0.15 * SR), np.float32)], [], int(0.15 * SR)
    for w in words:
        n = int(WORD_S * SR)
        f = 300.0 + 450.0 * w
        t = np.arange(n) / SR
        burst = (0.3 * np.sin(2 * np.pi * f * t) * np.hanning(n)).astype(np.float32)
        parts += [burst, np.zeros(int(GAP_S * SR), np.float32)]
        pos += n
        ends.append((w, pos))
        pos += int(GAP_S * SR)
    parts.append(np.zeros(int(0.2 * SR), np.float32))
    pcm = np.concatenate(parts) + 0.01 * rng.standard_normal(sum(len(p) for p in parts)).astype(np.float32)
    return pcm, ends


def frame_examples(pcm, ends, window, negative_label, rng):
    """WakeWordFrameBatchifier semantics: one window ending at each word's end (label = word), plus one random window
    labelled negative unless it ends within 45 ms of a word end (batchifier.py:74-110, simplified to known boundaries)."""
    out = []
    for w, e in ends:
        a = max(0, e - window)
        out.append((pcm[a:e], w))
    e = int(rng.integers(window // 2, len(pcm)))
    if all(abs(e - we) > 0.045 * SR for _, we in ends):
        out.append((pcm[max(0, e - window):e], negative_label))
    return out


def pad_batch(clips, window, device):
    """tensorize_audio_data(max_length=window, rand_append=True): zero padding on a random side (operator.py:89-109)."""
    audio = torch.zeros(len(clips), window)
    for i, c in enumerate(clips):
        c = torch.from_numpy(c)
        if random.random() < 0.5:
            audio[i, window - c.numel():] = c
        else:
            audio[i, : c.numel()] = c
    return audio.to(device)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", type=str, choices=RegisteredModel.registered_names(), default="res8")
    ap.add_argument("--workspace", type=str, default=str(Path("workspaces") / "default"))
    ap.add_argument("--load-weights", action="store_true")
    ap.add_argument("--load-last", action="store_true")
    ap.add_argument("--eval-freq", type=int, default=10)
    ap.add_argument("--eval", action="store_true")
    ap.add_argument("--synthetic", type=int, default=0, help="number of generated training clips")
    args = ap.parse_args(argv)
    if not args.synthetic:
        raise SystemExit("training.run.train on MI355X: dataset loading (howl/data/dataset) is outside the hot path; "
                         "pass --synthetic N to train on generated wake-word clips")

    use_frame = SETTINGS.training.objective == "frame"
    ctx = InferenceContext(SETTINGS.training.vocab, token_type=SETTINGS.training.token_type, use_blank=not use_frame)
    ws = Workspace(Path(args.workspace), delete_existing=not args.eval)
    writer = ws.summary_writer
    device = torch.device(SETTINGS.training.device)
    set_random_seed(SETTINGS.training.seed)
    rng = np.random.default_rng(SETTINGS.training.seed)
    n_vocab = len(SETTINGS.training.vocab)
    seq = list(SETTINGS.inference_engine.inference_sequence)
    window = int(SETTINGS.training.max_window_size_seconds * SR)

    def dataset(n, positive):
        clips = []
        for _ in range(n):
            words = list(seq) if positive else list(rng.permutation(n_vocab))[: int(rng.integers(1, n_vocab + 1))]
            if not positive and words == seq:
                words = words[::-1]
            clips.append(make_clip(words, n_vocab, rng))
        return clips

    train_clips = dataset(args.synthetic, True) + dataset(args.synthetic // 2, False)
    dev_pos, dev_neg = dataset(32, True), dataset(32, False)

    std_transform = StandardAudioTransform().to(device).eval()
    zmuv_transform = ZmuvTransform().to(device)
    model = RegisteredModel.find_registered_class(args.model)(ctx.num_labels).to(device).streaming()
    spectrogram_augmentations = (SpecAugmentTransform(),)      # train.py:277-278
    for pcm, _ in train_clips[:256]:
        zmuv_transform.update(std_transform(torch.from_numpy(pcm[:window * 4]).to(device)[None]))
    torch.save({k: v.cpu() for k, v in zmuv_transform.state_dict().items()}, str(ws.path / "zmuv.pt.bin"))
    if args.load_weights:
        ws.load_model(model, best=not args.load_last)
        model.to(device)

    def evaluate_engine(clips, prefix, positive, epoch):
        """train.py:42-94: run the engine over each clip, count detections."""
        std_transform.eval()
        model.eval()
        if use_frame:
            engine = FrameInferenceEngine(int(SETTINGS.training.max_window_size_seconds * 1000),
                                          int(SETTINGS.training.eval_stride_size_seconds * 1000), model, zmuv_transform, ctx)
        else:
            engine = InferenceEngine(model, zmuv_transform, ctx)
        tp = sum(int(bool(_infer(engine, pcm))) for pcm, _ in clips)
        n = len(clips)
        conf = dict(tp=tp, fn=n - tp, fp=0, tn=0) if positive else dict(tp=0, fn=0, fp=tp, tn=n - tp)
        with (ws.path / f"{engine.threshold}_results.csv").open("a") as f:
            csv.writer(f).writerow([prefix, epoch, conf["tp"], conf["tn"], conf["fp"], conf["fn"]])
        writer.add_scalar(f"{prefix}/Metric/tp_rate" if positive else f"{prefix}/Metric/fp_rate", tp / n, epoch)
        return conf

    def _infer(engine, pcm):
        engine.reset()
        model.streaming_state = None
        return engine.infer(torch.from_numpy(pcm).to(device))

    if args.eval:
        ws.load_model(model, best=not args.load_last)
        model.to(device)
        print(evaluate_engine(dev_pos, "Dev positive", True, 0), evaluate_engine(dev_neg, "Dev negative", False, 0))
        return

    ws.write_args(args)
    ws.save_settings(SETTINGS)
    params = [p for p in model.parameters() if p.requires_grad]
    fused = (use_frame and args.model in ("res8", "mobilenet")) or (not use_frame and args.model == "seq-lstm")
    if fused:
        trainer = FusedRes8Trainer(model, std_transform, zmuv_transform, SETTINGS.training.learning_rate,
                                   weight_decay=SETTINGS.training.weight_decay)
    else:
        optimizer = torch.optim.AdamW(params, SETTINGS.training.learning_rate, weight_decay=SETTINGS.training.weight_decay)
    criterion = torch.nn.CrossEntropyLoss()                        # frame objective; the CTC objective is ops.ctc_loss below
    B = SETTINGS.training.batch_size
    for epoch_idx in range(SETTINGS.training.num_epochs):
        std_transform.train()
        model.train()
        order = rng.permutation(len(train_clips))
        total_loss = torch.zeros((), device=device)
        for i in range(0, len(order) - B + 1, B):
            batch = [train_clips[j] for j in order[i:i + B]]
            if use_frame:
                ex = [e for pcm, ends in batch for e in frame_examples(pcm, ends, window, ctx.negative_label, rng)][:B * 4]
                audio = pad_batch([c for c, _ in ex], window, device)
                labels = torch.tensor([l for _, l in ex]).to(device)
                feats = std_transform.log_mel_for_model(audio, zmuv_transform)
                for aug in spectrogram_augmentations:
                    feats = aug(feats)
                if fused:
                    loss = trainer.step_on_features(feats, labels)
                else:
                    scores = model(feats, std_transform.compute_lengths(torch.full((len(ex),), window)))
                    optimizer.zero_grad()
                    loss = criterion(scores, labels)
                    loss.backward()
                    optimizer.step()
            else:
                batch = sorted(batch, key=lambda c: -len(c[0]))            # AudioSequenceBatchifier: longest first
                lmax = len(batch[0][0])
                audio = torch.zeros(len(batch), lmax)
                for k, (pcm, _) in enumerate(batch):
                    audio[k, : len(pcm)] = torch.from_numpy(pcm)
                lengths = std_transform.compute_lengths(torch.tensor([len(p) for p, _ in batch]))
                feats = std_transform.log_mel_for_model(audio.to(device), zmuv_transform)
                tl = torch.tensor([len(e) for _, e in batch])
                targets = torch.zeros(len(batch), max(1, int(tl.max())), dtype=torch.long)
                for k, (_, ends) in enumerate(batch):
                    targets[k, : len(ends)] = torch.tensor([w for w, _ in ends])
                # log_softmax + CTCLoss(blank) of train.py:291-296 as one fused kernel (loss and d loss / d scores)
                if fused:
                    loss = trainer.step_sequence_on_features(feats, lengths, targets, tl, ctx.blank_label, int(tl.max()))
                else:
                    scores = model(feats, lengths)
                    optimizer.zero_grad()
                    loss = ops.ctc_loss(scores, targets, lengths, tl, ctx.blank_label)
                    loss.backward()
                    optimizer.step()
            total_loss += loss.detach().reshape(())                       # accumulated on the device (train.py:303-304)
        if fused:
            trainer.decay_lr(SETTINGS.training.lr_decay)
        else:
            for group in optimizer.param_groups:
                group["lr"] *= SETTINGS.training.lr_decay
        writer.add_scalar("Training/Loss", total_loss / max(1, len(order) // B), epoch_idx)
        if epoch_idx % args.eval_freq == 0 and epoch_idx != 0:
            evaluate_engine(dev_pos, "Dev positive", True, epoch_idx)
            evaluate_engine(dev_neg, "Dev negative", False, epoch_idx)
        ws.save_model(model, best=False)
    pos = evaluate_engine(dev_pos, "Dev positive", True, SETTINGS.training.num_epochs)
    neg = evaluate_engine(dev_neg, "Dev negative", False, SETTINGS.training.num_epochs)
    ws.increment_model(model, pos["tp"] - neg["fp"])
    writer.close()
    print("dev positive:", pos, "dev negative:", neg)
    return pos, neg


if __name__ == "__main__":
    main()
