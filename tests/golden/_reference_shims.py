"""Import shims that let the reference's own hot-path modules run in this container.

Used ONLY by ``make_golden.py`` (run by hand in the authoring container, where /root/reference
exists) to capture golden vectors.  Nothing here travels into the product or runs in the test-suite.

What is shimmed and why:
* ``pydantic`` -> ``pydantic.v1``: the reference uses the v1 ``BaseSettings`` API.
* ``coloredlogs``, ``librosa``, ``torchvision.models``: absent here, only needed at import time.
* ``torchaudio.transforms``: absent here.  ``MelSpectrogram`` / ``ComputeDeltas`` are restated from the
  torchaudio-0.10.1 sources (Spectrogram -> torch.stft(...).abs().pow(2); MelScale -> matmul with the
  HTK filterbank; ComputeDeltas -> replicate-pad + grouped conv1d / 10).  The mel filterbank is taken
  from the reference's own vendored formula ``create_vtlp_fb_matrix(training=False)``
  (transform.py:373-410), which is the torchaudio formula.
Everything else (StandardAudioTransform, VtlpMelScale, ZmuvTransform, Res8, SimpleLstm,
SequentialLstm, InferenceContext, the engines, batchify, SpecAugment ...) is the reference's own code.
"""
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REFERENCE = "/root/reference"


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    import pydantic.v1 as pv1
    sys.modules["pydantic"] = pv1

    _module("coloredlogs", install=lambda *a, **k: None)
    eff = _module("librosa.effects", trim=None, time_stretch=None)
    core = _module("librosa.core", load=None)
    filt = _module("librosa.filters", mel=None, get_window=None)
    util = _module("librosa.util")
    _module("librosa", effects=eff, core=core, filters=filt, util=util)
    tvm = _module("torchvision.models", MobileNetV2=object, mobilenet_v2=None)
    _module("torchvision", models=tvm)
    _module("soundfile")

    class MelScale(nn.Module):
        def __init__(self, n_mels, sample_rate, f_min, f_max, n_stft):
            super().__init__()
            from howl.data.transform.transform import create_vtlp_fb_matrix
            self.register_buffer("fb", create_vtlp_fb_matrix(n_stft, f_min, f_max, n_mels, sample_rate, 1.0,
                                                             training=False))

        def forward(self, specgram):
            return torch.matmul(specgram.transpose(-1, -2), self.fb).transpose(-1, -2)

    class MelSpectrogram(nn.Module):
        def __init__(self, sample_rate=16000, n_fft=400, win_length=None, hop_length=None, f_min=0.0, f_max=None,
                     n_mels=128):
            super().__init__()
            self.sample_rate, self.n_fft, self.n_mels, self.f_min = sample_rate, n_fft, n_mels, f_min
            self.win_length = win_length if win_length is not None else n_fft
            self.hop_length = hop_length if hop_length is not None else self.win_length // 2
            self.f_max = f_max if f_max is not None else float(sample_rate // 2)
            self.register_buffer("window", torch.hann_window(self.win_length))
            self.mel_scale = MelScale(self.n_mels, self.sample_rate, self.f_min, self.f_max, self.n_fft // 2 + 1)

        def forward(self, waveform):
            spec = torch.stft(waveform, self.n_fft, hop_length=self.hop_length, win_length=self.win_length,
                              window=self.window, center=True, pad_mode="reflect", normalized=False, onesided=True,
                              return_complex=True)
            return self.mel_scale(spec.abs().pow(2.0))

    class ComputeDeltas(nn.Module):
        def __init__(self, win_length=5, mode="replicate"):
            super().__init__()
            self.win_length, self.mode = win_length, mode

        def forward(self, specgram):
            shape = specgram.size()
            x = specgram.reshape(1, -1, shape[-1])
            n = (self.win_length - 1) // 2
            denom = n * (n + 1) * (2 * n + 1) / 3
            x = F.pad(x, (n, n), mode=self.mode)
            kernel = torch.arange(-n, n + 1, 1, device=x.device, dtype=x.dtype).repeat(x.shape[1], 1, 1)
            return (F.conv1d(x, kernel, groups=x.shape[1]) / denom).reshape(shape)

    tat = _module("torchaudio.transforms", MelSpectrogram=MelSpectrogram, ComputeDeltas=ComputeDeltas)
    _module("torchaudio", transforms=tat)
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
