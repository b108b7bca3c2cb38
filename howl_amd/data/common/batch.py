"""Host containers feeding the hot path (``howl/data/common/batch.py:12-61``), same field names."""
from dataclasses import dataclass
from typing import Optional

import torch

__all__ = ["ClassificationBatch", "SequenceBatch"]


@dataclass
class ClassificationBatch:
    audio_data: torch.Tensor
    labels: Optional[torch.Tensor]
    lengths: torch.Tensor

    @classmethod
    def from_single(cls, audio_clip: torch.Tensor, label: int) -> "ClassificationBatch":
        return cls(audio_clip.unsqueeze(0), torch.tensor([label]), torch.tensor([audio_clip.size(-1)]))

    def pin_memory(self):
        self.audio_data = self.audio_data.pin_memory()
        if self.labels is not None:
            self.labels = self.labels.pin_memory()
        self.lengths = self.lengths.pin_memory()
        return self

    def to(self, device: torch.device) -> "ClassificationBatch":
        self.audio_data = self.audio_data.to(device)
        if self.labels is not None:
            self.labels = self.labels.to(device)
        self.lengths = self.lengths.to(device)
        return self


@dataclass
class SequenceBatch:
    audio_data: torch.Tensor
    labels: torch.Tensor
    audio_lengths: Optional[torch.Tensor]
    label_lengths: Optional[torch.Tensor]

    def to(self, device: torch.device) -> "SequenceBatch":
        self.audio_data = self.audio_data.to(device)
        self.labels = self.labels.to(device)
        if self.audio_lengths is not None:
            self.audio_lengths = self.audio_lengths.to(device)
        if self.label_lengths is not None:
            self.label_lengths = self.label_lengths.to(device)
        return self
