"""Minimal data plumbing for the entry points: a Google-Speech-Commands reader that needs no librosa (GSC clips are
16 kHz mono int16 wav) with the split / label rules of ``howl/data/dataset/gsc_dataset_loader.py:19-47``, a synthetic
stand-in for boxes without a dataset, and a device-resident clip bank with ``batchify`` semantics
(``howl/data/transform/operator.py:77-86``: sort by length descending, zero-pad right)."""
import json
import wave
from pathlib import Path
from types import SimpleNamespace
from typing import List, Tuple

import numpy as np
import torch

from howl_amd.data.common.batch import ClassificationBatch
from howl_amd.utils.synth import synthetic_pcm


def read_wav16k(path) -> torch.Tensor:
    with wave.open(str(path), "rb") as w:
        if w.getframerate() != 16000 or w.getnchannels() != 1 or w.getsampwidth() != 2:
            raise ValueError(f"{path}: expected 16 kHz mono int16 (resampling is outside the MI355X hot path)")
        return torch.from_numpy(np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").astype(np.float32) / 32768.0)


def load_gsc_splits(path: Path, vocab: List[str]):
    """-> (train, dev, test), each a list of (path, label); labels: index in vocab, else len(vocab)."""
    path = Path(path)
    split = {}
    for name, tag in (("testing_list.txt", "test"), ("validation_list.txt", "dev")):
        with (path / name).open() as f:
            split.update({k: tag for k in f.read().split("\n")})
    files = [p for p in sorted(path.glob("*/*.wav")) if "noise" not in str(p)]
    label_of = {k: i for i, k in enumerate(vocab)}
    out = {"train": [], "dev": [], "test": []}
    for p in files:
        key = str(Path(p.parent.name) / p.name)
        out[split.get(key, "train")].append((p, label_of.get(p.parent.name, len(vocab))))
    return out["train"], out["dev"], out["test"]


class ClipBank:
    """All clips of a split decoded once, truncated to ``max_len`` and kept on the device as one (N, max_len) matrix."""

    def __init__(self, clips: List[torch.Tensor], labels: List[int], max_len: int, device):
        n = len(clips)
        self.max_len = max_len
        audio = torch.zeros(n, max_len)
        lengths = torch.zeros(n, dtype=torch.long)
        for i, c in enumerate(clips):
            c = c[:max_len]                      # truncate_length (operator.py:73-74)
            audio[i, : c.numel()] = c
            lengths[i] = c.numel()
        self.audio = audio.to(device)
        self.lengths_host = lengths                      # batch maxima / sort orders are host decisions: no read-back per batch
        self.lengths = lengths.to(device)
        self.labels = torch.tensor(labels, dtype=torch.long).to(device)

    def __len__(self):
        return self.audio.size(0)

    def batch(self, idx: torch.Tensor) -> ClassificationBatch:
        """batchify: longest first, zero padded to the batch maximum."""
        idx_host = idx.cpu()
        lengths = self.lengths_host[idx_host]
        order = torch.argsort(lengths, descending=True, stable=True)
        idx_host, lengths = idx_host[order], lengths[order]
        lmax = int(lengths[0])     # the batch maximum, as batchify pads (res8's time-mean sees no extra pad frames)
        dev = self.audio.device
        idx = idx_host.to(dev)
        return ClassificationBatch(self.audio[idx, :lmax], self.labels[idx], lengths.to(dev))

    def index_batches(self, batch_size: int, shuffle: bool, drop_last: bool, generator=None):
        """Lists of clip ids per batch (for DeviceCollate, which gathers / augments / pads on the device)."""
        n = len(self)
        perm = (torch.randperm(n, generator=generator) if shuffle else torch.arange(n)).numpy()   # id arrays, not lists:
        end = n - (n % batch_size) if drop_last else n                                            # the collate indexes with them
        for i in range(0, end, batch_size):
            yield perm[i:i + batch_size]

    def batches(self, batch_size: int, shuffle: bool, drop_last: bool, generator=None):
        n = len(self)
        perm = torch.randperm(n, generator=generator) if shuffle else torch.arange(n)      # host: batch() sorts on the host
        end = n - (n % batch_size) if drop_last else n
        for i in range(0, end, batch_size):
            yield self.batch(perm[i:i + batch_size])


def synthetic_bank(n: int, max_len: int, num_labels: int, device, seed=0) -> ClipBank:
    pcm = synthetic_pcm(n, max_len, seed=seed)
    labels = [(i % 64) % num_labels for i in range(n)]     # the tone frequency of clip i encodes its label
    return ClipBank([pcm[i] for i in range(n)], labels, max_len, device)


# ---- wake-word datasets (training.run.train) --------------------------------------------------------------------------
def load_howl_splits(path: Path, prefix: str = "aligned-"):
    """A Howl-format dataset directory (``howl/data/dataset/dataset_loader.py:34-70``, ``WakeWordDatasetLoader``):
    ``<prefix>metadata-{training,dev,test}.jsonl`` with one JSON object per line (``path``, ``transcription``,
    ``end_timestamps`` per character in ms) and the clips under ``audio/``.  -> three lists of metadata records."""
    path = Path(path)

    def load(name):
        out = []
        with (path / f"{prefix}metadata-{name}.jsonl").open() as f:
            for line in f:
                if line.strip():
                    d = json.loads(line)
                    out.append(SimpleNamespace(path=(path / "audio" / d["path"]).absolute(), transcription=d["transcription"],
                                               end_timestamps=d.get("end_timestamps")))
        return out

    return load("training"), load("dev"), load("test")


class WakeWordClipBank:
    """Clips of a wake-word split decoded once and kept on the device RAGGED: one flat sample buffer, clip i at
    ``offsets[i] .. offsets[i] + lengths[i]`` (each start 16-byte aligned), so a single long negative costs its own length
    and not a row of that width for every clip.  The collate / gather kernels address a clip as ``bank + idx * bank_ld``: the
    bank is handed to them as the (total, 1) view ``rows`` (``bank_ld`` = 1) with ``idx`` = the clip's offset
    (``DeviceCollate(row_offsets=...)``).  Each clip comes with the descriptor the batchifiers work on (``DeviceClip``: clip
    id, length, frame labels from the context's labeler, transcription)."""

    MAX_OFFSET = (1 << 31) - 1     # offsets travel as int32 (8 GiB of samples, ~37 h of 16 kHz audio per split)

    def __init__(self, clips: List[torch.Tensor], metadata: list, labeler, device):
        from howl_amd.data.transform.batchifier import DeviceClip
        lengths = [int(c.numel()) for c in clips]
        offsets, total = [], 0
        for n in lengths:
            offsets.append(total)
            total += (n + 3) & ~3
        if total > self.MAX_OFFSET:
            raise MemoryError(f"clip bank of {total} samples ({total * 4 / 2**30:.0f} GiB) exceeds the 32-bit offsets of the collate "
                              f"kernels: shard the split")
        flat = torch.zeros(max(total, 1))
        for c, o, n in zip(clips, offsets, lengths):
            flat[o:o + n] = c
        self.flat = flat.to(device)
        self.rows = self.flat.unsqueeze(1)          # (total, 1): row stride 1, "row" index = sample offset
        self.offsets = offsets
        self.max_len = max(lengths) if lengths else 0
        self.lengths = torch.tensor(lengths)
        self.examples = []
        for i, (n, m) in enumerate(zip(lengths, metadata)):
            tl = labeler.compute_frame_labels(m).timestamp_label_map if m.end_timestamps is not None else {}
            self.examples.append(DeviceClip(i, n, tl, m.transcription))

    def __len__(self):
        return len(self.examples)

    def clip(self, i: int) -> torch.Tensor:
        return self.flat[self.offsets[i]: self.offsets[i] + self.examples[i].num_samples]

    def subset(self, keep) -> List[int]:
        return [i for i, ex in enumerate(self.examples) if keep(ex)]
