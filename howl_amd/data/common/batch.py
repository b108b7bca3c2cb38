"""Batch containers handed to the hot path; field names and methods follow ``howl/data/common/batch.py:12-61`` so that
collate functions and training loops written against the reference keep working."""
from dataclasses import dataclass, fields
from typing import Optional

import torch

__all__ = ["ClassificationBatch", "SequenceBatch"]


class _TensorBundle:
    """Moves / pins every tensor-valued dataclass field in place (``None`` fields are skipped) and returns ``self``."""

    def _map_tensors(self, fn):
        for f in fields(self):
            value = getattr(self, f.name)
            if torch.is_tensor(value):
                setattr(self, f.name, fn(value))
        return self

    def to(self, device):
        return self._map_tensors(lambda t: t.to(device))

    def pin_memory(self):
        return self._map_tensors(lambda t: t.pin_memory())


@dataclass
class ClassificationBatch(_TensorBundle):
    audio_data: torch.Tensor                 # (B, Lmax) right-padded waveforms
    labels: Optional[torch.Tensor]           # (B,) class indices
    lengths: torch.Tensor                    # (B,) valid samples per row

    @classmethod
    def from_single(cls, audio_clip: torch.Tensor, label: int) -> "ClassificationBatch":
        n = audio_clip.size(-1)
        return cls(audio_data=audio_clip[None], labels=torch.tensor([label]), lengths=torch.tensor([n]))


@dataclass
class SequenceBatch(_TensorBundle):
    audio_data: torch.Tensor                 # (B, Lmax)
    labels: torch.Tensor                     # (B, Smax) token ids
    audio_lengths: Optional[torch.Tensor]
    label_lengths: Optional[torch.Tensor]
