"""Device-side collate for device-resident clips: the reference's training collate chain
``compose(truncate_length, TimeshiftTransform().train(), NoiseTransform().train(), batchify)``
(``training/run/pretrain_gsc.py:78-80``; ``howl/data/transform/transform.py:120-196``, ``operator.py:73-86``) as one
``howl_collate_augment`` launch per batch -- optionally with ``DatasetMixer`` (``transform.py:199-231``) in front, as
``training/run/train.py:218`` composes it when a noise dataset is configured.

The per-batch gates and per-sample magnitudes are drawn on the host from the same ``random`` generator in the same order
as the reference modules draw them (one gate draw per augmentation parameter per batch; per example: shift amount then
head/tail for Timeshift, strength for each active Noise parameter); only the noise *samples* differ (counter-based
device generator instead of torch's CPU generator), so augmented batches match the reference in distribution.
"""
import random

import torch

from howl_amd import ops
from howl_amd.data.common.batch import ClassificationBatch

__all__ = ["DeviceCollate"]

TIMESHIFT_DOMAIN, TIMESHIFT_IDX, TIMESHIFT_PROB = [0.25, 0.5, 0.75, 1], 0, 0.75
WHITE_DOMAIN, WHITE_IDX = [0.0001, 0.00025, 0.0005, 0.001, 0.002], 3
SP_DOMAIN, SP_IDX = [1 / 20000, 1 / 15000, 1 / 10000, 1 / 5000, 1 / 2500], 2
NOISE_PROB = 0.75
MIXER_DOMAIN, MIXER_IDX, MIXER_PROB = [0.1, 0.2, 0.3, 0.4, 0.5], 1, 0.75


class DeviceCollate:
    def __init__(self, bank_audio, bank_lengths, bank_labels, max_len: int, sr: int = 16000, seed: int = None,
                 training: bool = True, background=None, do_replace: bool = False):
        """``background`` = (bg_audio (N, Lbg) on the device, bg_lengths): the noise dataset of ``DatasetMixer``."""
        self.audio, self.lengths, self.labels = bank_audio, bank_lengths.tolist(), bank_labels
        self.bg_audio = None if background is None else background[0]
        self.bg_lengths = None if background is None else [int(v) for v in background[1]]
        self.do_replace = do_replace
        self.max_len, self.sr, self.training = max_len, sr, training
        self.rand = random if seed is None else random.Random(seed)
        self._calls = 0
        self._seed = 0 if seed is None else seed

    def draw(self, clip_ids):
        """Host-side parameter draws in the reference's order; returns per-sample lists (before the length sort)."""
        n = len(clip_ids)
        lens = [min(self.lengths[i], self.max_len) for i in clip_ids]            # truncate_length
        self.last_mix = self.draw_mixer(lens)                                     # DatasetMixer comes first (train.py:218)
        shift, head = [0] * n, [0] * n
        if self.rand.random() < TIMESHIFT_PROB and self.training:                 # AugmentModule.forward gate
            for k in range(n):
                w = min(int(self.rand.random() * TIMESHIFT_DOMAIN[TIMESHIFT_IDX] * self.sr), int(0.5 * lens[k]))
                shift[k] = w
                head[k] = 1 if self.rand.random() < 0.5 else 0
        sigma, sp = [0.0] * n, [0.0] * n
        if self.rand.random() < NOISE_PROB and self.training:                     # "white" parameter
            for k in range(n):
                sigma[k] = WHITE_DOMAIN[WHITE_IDX] * self.rand.random()
        if self.rand.random() < NOISE_PROB and self.training:                     # "salt_pepper" parameter
            for k in range(n):
                sp[k] = SP_DOMAIN[SP_IDX] * self.rand.random()
        return lens, shift, head, sigma, sp

    def draw_mixer(self, lens):
        """``DatasetMixer.forward`` draws (AugmentModule gate per parameter, then per example choice / randint / alpha)."""
        if self.bg_audio is None:
            return None
        n = len(lens)
        bg_id, bg_off, alpha = [0] * n, [0] * n, [0.0] * n
        for name, prob in (("strength", MIXER_PROB), ("replace", 0.1 if self.do_replace else 0.0)):
            if self.rand.random() < prob and self.training:
                for k in range(n):
                    j = self.rand.choice(range(len(self.bg_lengths)))
                    while self.bg_lengths[j] < lens[k]:
                        j = self.rand.choice(range(len(self.bg_lengths)))
                    b = self.rand.randint(lens[k], self.bg_lengths[j])
                    a = 1.0 if name == "replace" else self.rand.random() * MIXER_DOMAIN[MIXER_IDX]
                    bg_id[k], bg_off[k], alpha[k] = j, b - lens[k], a   # "replace" (alpha 1) overrides an earlier mix
        return bg_id, bg_off, alpha

    def __call__(self, clip_ids) -> ClassificationBatch:
        clip_ids = list(clip_ids)
        lens, shift, head, sigma, sp = self.draw(clip_ids)
        out_len = [l - w for l, w in zip(lens, shift)]
        order = sorted(range(len(clip_ids)), key=lambda k: -out_len[k])          # batchify: longest first (stable)
        pick = lambda a: [a[k] for k in order]
        dev = self.audio.device
        i32 = lambda a: torch.tensor(pick(a), dtype=torch.int32).to(dev, non_blocking=True)
        f32 = lambda a: torch.tensor(pick(a), dtype=torch.float32).to(dev, non_blocking=True)
        lmax = max(out_len)
        self._calls += 1
        mix = None
        if self.last_mix is not None and any(a != 0.0 for a in self.last_mix[2]):
            mix = (self.bg_audio, i32(self.last_mix[0]), i32(self.last_mix[1]), f32(self.last_mix[2]))
        audio = ops.collate_augment(self.audio, i32(clip_ids), i32(lens), i32(shift), i32(head), f32(sigma), f32(sp),
                                    (self._seed << 32) ^ self._calls, lmax, mix=mix)
        idx = torch.tensor(pick(clip_ids), dtype=torch.long).to(dev, non_blocking=True)
        return ClassificationBatch(audio, self.labels[idx], torch.tensor(pick(out_len)).to(dev, non_blocking=True))
