"""``python -m training.run.train --model M --workspace W -i DS... [--load-weights --load-last --eval-freq N --eval]`` on MI355X.

The training / evaluation flow of ``training/run/train.py:35-324`` of the reference for the two objectives its presets
use (``envs/res8.env``: frame-level cross-entropy; ``envs/seq-lstm.env``: CTC): InferenceContext label space, ZMUV pass,
``model.streaming()``, per step  collate -> frontend -> ZMUV -> SpecAugment -> model -> loss -> AdamW, per-epoch LR decay and
``streaming_state`` reset, evaluation every ``--eval-freq`` epochs through ``FrameInferenceEngine`` / ``InferenceEngine`` with
TP/FN (positive clips) and FP/TN (negative clips) counts written to ``<threshold>_results.csv`` like ``train.py:66-94``.

What is different is where the work happens.  The reference decodes and collates in DataLoader worker processes; here every
split is decoded once into a device-resident clip bank (``WakeWordClipBank``) and the collate chain of ``train.py:211-229``
-- ``[DatasetMixer,] TimeshiftTransform, NoiseTransform, WakeWordFrameBatchifier | AudioSequenceBatchifier`` -- is a set of
host decisions (the reference's draws, in its order) followed by ONE device launch per batch (``DeviceCollate``).
``TimestretchTransform`` (librosa phase vocoder) is outside the hot path and not applied.

Datasets: ``-i`` takes Howl-format dataset directories (``aligned-metadata-{training,dev,test}.jsonl`` + ``audio/*.wav``,
16 kHz mono int16 -- resampling / other codecs are the reference's librosa path and out of scope); ``--synthetic N``
generates wake-word clips instead (word w of the vocabulary is a tone burst of its own pitch; positives are the wake
sequence, negatives shuffled / partial sequences) in the same metadata form, so both go through the same labeler.
"""
import argparse
import csv
import logging
import random
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

from howl_amd import ops, parallel
from howl_amd.context import InferenceContext
from howl_amd.data.collate import DeviceCollate
from howl_amd.data.common.tokenizer import WakeWordTokenizer
from howl_amd.data.transform.batchifier import AudioSequenceBatchifier, WakeWordFrameBatchifier
from howl_amd.data.transform.operator import ZmuvTransform
from howl_amd.data.transform.transform import SpecAugmentTransform, StandardAudioTransform
from howl_amd.model import RegisteredModel
from howl_amd.model.cnn import require_supported_mels
from howl_amd.model.inference import FrameInferenceEngine, InferenceEngine
from howl_amd.settings import SETTINGS
from howl_amd.training.data import WakeWordClipBank, load_howl_splits, read_wav16k
from howl_amd.training.fused import FusedTrainer
from howl_amd.utils.random_utils import set_random_seed
from howl_amd.workspace import Workspace

SR = 16000
WORD_S = 0.3     # seconds per word burst
GAP_S = 0.08


def make_clip(word_ids, vocab, rng):
    """Tone bursts for ``word_ids``; returns (pcm, metadata) with per-character end timestamps (ms) like an aligned
    dataset record: the characters of a word are spread evenly over its burst, the space after it ends with the word."""
    lead = int(0.15 * SR)
    parts, stamps, pos = [np.zeros(lead, np.float32)], [], lead
    words = [vocab[w] for w in word_ids]
    for k, (w, text) in enumerate(zip(word_ids, words)):
        n = int(WORD_S * SR)
        t = np.arange(n) / SR
        parts += [(0.3 * np.sin(2 * np.pi * (300.0 + 450.0 * w) * t) * np.hanning(n)).astype(np.float32),
                  np.zeros(int(GAP_S * SR), np.float32)]
        start_ms, end_ms = pos / SR * 1000, (pos + n) / SR * 1000
        stamps += [start_ms + (end_ms - start_ms) * (c + 1) / len(text) for c in range(len(text))]
        if k + 1 < len(words):
            stamps.append(end_ms)
        pos += n + int(GAP_S * SR)
    parts.append(np.zeros(int(0.2 * SR), np.float32))
    pcm = np.concatenate(parts)
    pcm = pcm + 0.01 * rng.standard_normal(pcm.size).astype(np.float32)
    return torch.from_numpy(pcm), SimpleNamespace(path=None, transcription=" ".join(words), end_timestamps=stamps)


def synthetic_split(n, positive, vocab, seq, rng):
    clips, meta = [], []
    for _ in range(n):
        ids = list(seq) if positive else [int(v) for v in rng.permutation(len(vocab))[: int(rng.integers(1, len(vocab) + 1))]]
        if not positive and ids == list(seq):
            ids = ids[::-1]
        c, m = make_clip(ids, vocab, rng)
        clips.append(c)
        meta.append(m)
    return clips, meta


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", type=str, choices=RegisteredModel.registered_names(), default="res8")
    ap.add_argument("--workspace", type=str, default=str(Path("workspaces") / "default"))
    ap.add_argument("--load-weights", action="store_true")
    ap.add_argument("--load-last", action="store_true")
    ap.add_argument("--dataset-paths", "-i", type=str, nargs="+", default=[])
    ap.add_argument("--eval-freq", type=int, default=10)
    ap.add_argument("--eval", action="store_true")
    ap.add_argument("--synthetic", type=int, default=0, help="number of generated positive training clips (no -i needed)")
    args = ap.parse_args(argv)
    if not args.synthetic and not args.dataset_paths:
        raise SystemExit("training.run.train: give Howl-format dataset directories with -i, or --synthetic N")

    use_frame = SETTINGS.training.objective == "frame"
    set_random_seed(SETTINGS.training.seed)
    # data parallel (one process per GPU under torch.distributed.run): BATCH_SIZE is the global batch, every rank takes its
    # contiguous share of each (identically drawn) batch, the fused steps sum the flat gradient over RCCL, rank 0 writes
    device = torch.device(SETTINGS.training.device)
    rank, world, device = parallel.init_from_env(device)
    main_rank = rank == 0
    zmuv_on_disk = (Path(args.workspace) / "zmuv.pt.bin").exists()      # asked before rank 0 writes anything
    if main_rank:
        ws = Workspace(Path(args.workspace), delete_existing=not args.eval)
    parallel.barrier()
    if not main_rank:
        ws = Workspace(Path(args.workspace), delete_existing=False, writable=False)
    writer = ws.summary_writer
    ctx = InferenceContext(SETTINGS.training.vocab, token_type=SETTINGS.training.token_type, use_blank=not use_frame)
    rng = np.random.default_rng(SETTINGS.training.seed)
    vocab = list(SETTINGS.training.vocab)
    seq = list(SETTINGS.inference_engine.inference_sequence)

    # ---- datasets -> device clip banks ----------------------------------------------------------------------------
    splits = {"training": ([], []), "dev": ([], []), "test": ([], [])}
    for ds_path in args.dataset_paths:
        for name, records in zip(("training", "dev", "test"), load_howl_splits(Path(ds_path))):
            for m in records:
                splits[name][0].append(read_wav16k(m.path))
                splits[name][1].append(m)
    if args.synthetic:
        for name, n_pos, n_neg in (("training", args.synthetic, args.synthetic // 2), ("dev", 32, 32)):
            for positive, n in ((True, n_pos), (False, n_neg)):
                c, m = synthetic_split(n, positive, vocab, seq, rng)
                splits[name][0].extend(c)
                splits[name][1].extend(m)
    train_bank = WakeWordClipBank(*splits["training"], ctx.labeler, device)
    if splits["dev"][0]:
        dev_bank = WakeWordClipBank(*splits["dev"], ctx.labeler, device)
    else:
        logging.warning("no dev split in the datasets: the engine evaluation (results CSV, best-model score) runs on TRAINING clips")
        dev_bank = train_bank
    is_pos = lambda ex: ctx.searcher.search(ex.transcription)      # train.py:176-183
    dev_pos, dev_neg = dev_bank.subset(is_pos), dev_bank.subset(lambda ex: not is_pos(ex))

    # ---- collate chain: host draws + one launch per batch (train.py:196-229) ------------------------------------------------
    window_ms = int(SETTINGS.training.max_window_size_seconds * 1000)
    if use_frame:
        batchifier = WakeWordFrameBatchifier(ctx.negative_label, window_size_ms=window_ms)
    else:
        batchifier = AudioSequenceBatchifier(ctx.negative_label, WakeWordTokenizer(ctx.vocab, ignore_oov=False))
    collate = DeviceCollate(train_bank.rows, train_bank.lengths, None, max_len=train_bank.max_len, row_offsets=train_bank.offsets,
                            seed=(SETTINGS.training.seed ^ 0x5DEECE66D) if world > 1 else None, replica=rank)

    std_transform = StandardAudioTransform().to(device).eval()
    zmuv_transform = ZmuvTransform().to(device)
    model = RegisteredModel.find_registered_class(args.model)(ctx.num_labels).to(device).streaming()
    require_supported_mels(model)      # res8 with NUM_MELS other than 40 / 80: an error here, not at the first batch
    spectrogram_augmentations = (SpecAugmentTransform().train(),)      # train.py:277-278
    if zmuv_on_disk:
        zmuv_transform.load_state_dict(torch.load(str(ws.path / "zmuv.pt.bin")))
        zmuv_transform.to(device)
    else:
        # every rank makes the same pass (same clips, same kernels, same order: `rng` stays in step across the ranks);
        # rank 0's statistics are broadcast anyway
        for i in rng.permutation(len(train_bank))[:2001]:              # prep_dl: single shuffled examples (train.py:231-245)
            zmuv_transform.update(std_transform(train_bank.clip(int(i))[None]))
        parallel.broadcast_([zmuv_transform.total, zmuv_transform.mean, zmuv_transform.mean2])
        if main_rank:   # only freshly computed statistics are written: the other ranks may still be reading an existing file
            torch.save({k: v.cpu() for k, v in zmuv_transform.state_dict().items()}, str(ws.path / "zmuv.pt.bin"))
    if args.load_weights:
        ws.load_model(model, best=not args.load_last)
        model.to(device)

    def evaluate_engine(bank, ids, prefix, positive, epoch):
        """train.py:42-94: run the engine over each clip, count detections."""
        std_transform.eval()
        model.eval()
        if use_frame:
            engine = FrameInferenceEngine(window_ms, int(SETTINGS.training.eval_stride_size_seconds * 1000), model,
                                          zmuv_transform, ctx)
        else:
            engine = InferenceEngine(model, zmuv_transform, ctx)
        hits = 0
        mine = ids[rank::world]            # clips dealt round-robin to the ranks, detections summed
        if use_frame:                      # all windows of 64 clips at a time in one batch (FrameInferenceEngine.infer_many)
            for lo in range(0, len(mine), 64):
                model.streaming_state = None
                hits += sum(int(h) for h in engine.infer_many([bank.clip(i) for i in mine[lo:lo + 64]]))
            mine = []
        for i in mine:
            engine.reset()
            model.streaming_state = None
            hits += int(bool(engine.infer(bank.clip(i))))
        model.streaming_state = None       # whatever the last clip left behind must not leak into training batches
        hits = int(parallel.allreduce_scalars_(torch.tensor([float(hits)], device=device)).item())
        n = len(ids)
        conf = dict(tp=hits, fn=n - hits, fp=0, tn=0) if positive else dict(tp=0, fn=0, fp=hits, tn=n - hits)
        if main_rank:
            with (ws.path / f"{engine.threshold}_results.csv").open("a") as f:
                csv.writer(f).writerow([prefix, epoch, conf["tp"], conf["tn"], conf["fp"], conf["fn"]])
        writer.add_scalar(f"{prefix}/Metric/tp_rate" if positive else f"{prefix}/Metric/fp_rate", hits / max(n, 1), epoch)
        return conf

    if args.eval:
        ws.load_model(model, best=not args.load_last)
        model.to(device)
        pos, neg = evaluate_engine(dev_bank, dev_pos, "Dev positive", True, 0), evaluate_engine(dev_bank, dev_neg, "Dev negative", False, 0)
        if main_rank:
            print(pos, neg)
        return pos, neg

    ws.write_args(args)
    ws.save_settings(SETTINGS)
    params = [p for p in model.parameters() if p.requires_grad]
    fused = (use_frame and args.model in ("res8", "mobilenet", "lstm")) or (not use_frame and args.model == "seq-lstm")
    needs_lengths = getattr(model, "NEEDS_LENGTHS", False)
    if fused:
        trainer = FusedTrainer(model, std_transform, zmuv_transform, SETTINGS.training.learning_rate,
                               weight_decay=SETTINGS.training.weight_decay)
        trainer.broadcast_parameters()                             # rank 0's initial weights and BatchNorm buffers
    elif world > 1:
        raise NotImplementedError(f"{args.model} with the {'frame' if use_frame else 'ctc'} objective has no fused step: "
                                  "data-parallel training needs one (res8 / mobilenet / lstm: frame, seq-lstm: ctc)")
    else:
        optimizer = torch.optim.AdamW(params, SETTINGS.training.learning_rate, weight_decay=SETTINGS.training.weight_decay)
    criterion = torch.nn.CrossEntropyLoss()                        # frame objective; the CTC objective is ops.ctc_loss below
    B = SETTINGS.training.batch_size
    for epoch_idx in range(SETTINGS.training.num_epochs):
        std_transform.train()
        model.train()
        model.streaming_state = None                               # train.py:284
        order = [int(v) for v in rng.permutation(len(train_bank))]
        total_loss = torch.zeros((), device=device)
        n_batches = 0
        for i in range(0, len(order), B):
            if len(order) - i < world:
                continue                                           # a ragged last batch smaller than the world size: every rank skips it
            examples = parallel.shard([train_bank.examples[j] for j in order[i:i + B]])
            if use_frame:
                batch = collate.frame_batch(examples, batchifier)
                frame_lengths = std_transform.compute_lengths(batch.lengths)
                feats = std_transform.log_mel_for_model(batch.audio_data, zmuv_transform)
                for aug in spectrogram_augmentations:              # after ZMUV, for both objectives (train.py:289-290)
                    feats = aug(feats)
                if fused and needs_lengths:
                    loss = trainer.step_on_features(feats, batch.labels, frame_lengths)
                elif fused:
                    loss = trainer.step_on_features(feats, batch.labels)
                else:
                    scores = model(feats, frame_lengths)
                    optimizer.zero_grad()
                    loss = criterion(scores, batch.labels)
                    loss.backward()
                    optimizer.step()
            else:
                batch = collate.sequence_batch(examples, batchifier)
                frame_lengths = std_transform.compute_lengths(batch.audio_lengths)
                feats = std_transform.log_mel_for_model(batch.audio_data, zmuv_transform)
                for aug in spectrogram_augmentations:
                    feats = aug(feats)
                if model.streaming_state is not None and model.streaming_state[0].shape[1] != feats.shape[0]:
                    model.streaming_state = None                   # a ragged last batch cannot inherit the carried state
                max_target = int(batch.label_lengths.max())
                # log_softmax + CTCLoss(blank) of train.py:291-296 as one fused kernel (loss and d loss / d scores)
                if fused:
                    loss = trainer.step_sequence_on_features(feats, frame_lengths, batch.labels, batch.label_lengths,
                                                             ctx.blank_label, max_target)
                else:
                    scores = model(feats, frame_lengths)
                    optimizer.zero_grad()
                    loss = ops.ctc_loss(scores, batch.labels, frame_lengths, batch.label_lengths, ctx.blank_label)
                    loss.backward()
                    optimizer.step()
            total_loss += loss.detach().reshape(())                       # accumulated on the device (train.py:303-304)
            n_batches += 1
        if fused:
            trainer.decay_lr(SETTINGS.training.lr_decay)
        else:
            for group in optimizer.param_groups:
                group["lr"] *= SETTINGS.training.lr_decay
        writer.add_scalar("Training/Loss", total_loss / max(1, n_batches), epoch_idx)
        if epoch_idx % args.eval_freq == 0 and epoch_idx != 0:
            evaluate_engine(dev_bank, dev_pos, "Dev positive", True, epoch_idx)
            evaluate_engine(dev_bank, dev_neg, "Dev negative", False, epoch_idx)
        ws.save_model(model, best=False)
    pos = evaluate_engine(dev_bank, dev_pos, "Dev positive", True, SETTINGS.training.num_epochs)
    neg = evaluate_engine(dev_bank, dev_neg, "Dev negative", False, SETTINGS.training.num_epochs)
    ws.increment_model(model, pos["tp"] - neg["fp"])
    writer.close()
    parallel.barrier()
    if main_rank:
        print("dev positive:", pos, "dev negative:", neg)
    return pos, neg


if __name__ == "__main__":
    main()
