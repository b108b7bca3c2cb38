"""Oracle: audio frontend (STFT -> mel -> log -> deltas) and ZMUV.  Test infrastructure only.

Restates ``howl/data/transform/transform.py:234-296`` (StandardAudioTransform),
``:373-410`` (create_vtlp_fb_matrix), ``howl/data/transform/operator.py:119-146``
(ZmuvTransform) and the torchaudio-0.10 ``Spectrogram`` / ``MelScale`` /
``ComputeDeltas`` arithmetic those call.
"""
import math

import torch
import torch.nn.functional as F

SAMPLE_RATE = 16000
N_FFT = 512
HOP = 200
N_FREQS = N_FFT // 2 + 1
LOG_EPS = 1e-7


def mel_fb(n_mels: int = 40, alpha: float = None, sample_rate: int = SAMPLE_RATE, f_hi: int = 4800) -> torch.Tensor:
    """(257, n_mels) HTK mel triangles; ``alpha`` != None applies the VTLP warp.

    Follows ``transform.py:373-410`` statement for statement, including the quirk at
    :397-401 that the ``>`` mask is re-evaluated on the already-scaled tensor.
    ``alpha=None`` is the ``training=False`` branch == torchaudio's ``melscale_fbanks``
    (f_min=0, f_max=sample_rate//2, norm=None, htk).
    """
    S = sample_rate
    f_min, f_max = 0.0, float(sample_rate // 2)
    all_freqs = torch.linspace(0, sample_rate // 2, N_FREQS)
    m_min = 2595.0 * math.log10(1.0 + (f_min / 700.0))
    m_max = 2595.0 * math.log10(1.0 + (f_max / 700.0))
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    if alpha is not None:
        thr = f_hi * min(alpha, 1) / alpha
        f_pts[f_pts <= thr] *= alpha
        f = f_pts[f_pts > thr]
        f_pts[f_pts > thr] = S / 2 - ((S / 2 - f_hi * min(alpha, 1)) / (S / 2 - thr)) * (S / 2 - f)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    zero = torch.zeros(1)
    down_slopes = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up_slopes = slopes[:, 2:] / f_diff[1:]
    return torch.max(zero, torch.min(down_slopes, up_slopes))


def power_spectrogram(audio: torch.Tensor) -> torch.Tensor:
    """(B, L) -> (B, 257, T), T = 1 + L // 200.  torchaudio ``Spectrogram(power=2)`` as called at
    ``transform.py:249-254``: centred, reflect-padded, periodic Hann, unnormalised, one-sided."""
    window = torch.hann_window(N_FFT, periodic=True, dtype=audio.dtype, device=audio.device)
    spec = torch.stft(audio, N_FFT, hop_length=HOP, win_length=N_FFT, window=window, center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    return spec.abs().pow(2.0)


def mel_spectrogram(audio: torch.Tensor, fb: torch.Tensor) -> torch.Tensor:
    """(B, L) -> (B, n_mels, T).  torchaudio ``MelScale.forward`` == ``transform.py:446``."""
    spec = power_spectrogram(audio)
    return torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)


def compute_deltas(x: torch.Tensor, win_length: int = 5) -> torch.Tensor:
    """torchaudio ``ComputeDeltas(win_length=5, mode='replicate')`` (call sites ``transform.py:264,278-279``)."""
    shape = x.size()
    x = x.reshape(1, -1, shape[-1])
    n = (win_length - 1) // 2
    denom = n * (n + 1) * (2 * n + 1) / 3
    x = F.pad(x, (n, n), mode="replicate")
    kernel = torch.arange(-n, n + 1, 1, dtype=x.dtype).repeat(x.shape[1], 1, 1)
    return (F.conv1d(x, kernel, groups=x.shape[1]) / denom).reshape(shape)


def standard_audio_transform(audio: torch.Tensor, fb: torch.Tensor, mels_only: bool = False,
                             deltas_only: bool = False) -> torch.Tensor:
    """``StandardAudioTransform._execute_op`` (``transform.py:271-280``): (B, L) -> (B, 3, M, T)."""
    log_mels = audio if deltas_only else mel_spectrogram(audio, fb).add_(LOG_EPS).log_().contiguous()
    if mels_only:
        return log_mels
    deltas = compute_deltas(log_mels)
    accels = compute_deltas(deltas)
    return torch.stack((log_mels, deltas, accels), 1)


def compute_lengths(length: torch.Tensor) -> torch.Tensor:
    """``transform.py:290-296``: floor((len - win) / hop) + 1, ignoring the centre padding."""
    return (torch.div(length - N_FFT, HOP, rounding_mode="floor") + 1).long()


class Zmuv:
    """``ZmuvTransform`` (``operator.py:119-146``): running scalar mean / mean-of-squares."""

    def __init__(self):
        self.total = torch.zeros(1)
        self.mean = torch.zeros(1)
        self.mean2 = torch.zeros(1)

    def update(self, data: torch.Tensor, mask: torch.Tensor = None):
        if mask is not None:
            data = data * mask
            mask_size = mask.sum().item()
        else:
            mask_size = data.numel()
        self.mean = (data.sum() + self.mean * self.total) / (self.total + mask_size)
        self.mean2 = ((data ** 2).sum() + self.mean2 * self.total) / (self.total + mask_size)
        self.total = self.total + mask_size

    @property
    def std(self):
        return (self.mean2 - self.mean ** 2).sqrt()

    def __call__(self, x):
        return (x - self.mean) / self.std


def spec_augment_apply(x: torch.Tensor, f0, f, t0, t) -> torch.Tensor:
    """Mask application of ``SpecAugmentTransform.fmask/tmask`` (``transform.py:309-327``) with the
    drawn parameters given explicitly (per sample; t[i] < 0 means 'skipped')."""
    x = x.clone()
    for i in range(x.size(0)):
        if f[i] >= 0:
            x[i, :, f0[i]:f0[i] + f[i]] = 0
        if t[i] >= 0:
            x[i, :, :, t0[i]:t0[i] + t[i]] = 0
    return x


MIXER_STRENGTH_DOMAIN, MIXER_STRENGTH_IDX, MIXER_PROB = [0.1, 0.2, 0.3, 0.4, 0.5], 1, 0.75


def dataset_mixer(rand, waveforms, backgrounds, training: bool = True, do_replace: bool = False):
    """``DatasetMixer.forward`` (``transform.py:90-97,199-231``) with ``rand`` a ``random.Random``-like stream: one gate draw
    per augmentation parameter ("strength" prob 0.75, "replace" prob 0.1 if do_replace else 0 -- the draw is consumed
    either way); when "strength" fires, per example: background = rand.choice (redrawn while shorter than the waveform),
    b = rand.randint(len, bg_len), alpha = rand.random() * 0.2, out = wf * (1 - alpha) + bg[b - len:b] * alpha."""
    out = list(waveforms)
    for name, prob in (("strength", MIXER_PROB), ("replace", 0.1 if do_replace else 0.0)):
        if rand.random() < prob and training:
            mixed = []
            for wf in out:
                n = wf.numel()
                bg = rand.choice(backgrounds)
                while bg.numel() < n:
                    bg = rand.choice(backgrounds)
                b = rand.randint(n, bg.numel())
                alpha = 1.0 if name == "replace" else rand.random() * MIXER_STRENGTH_DOMAIN[MIXER_STRENGTH_IDX]
                mixed.append(wf * (1 - alpha) + bg[b - n:b] * alpha)
            out = mixed
    return out
