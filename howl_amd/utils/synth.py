"""Deterministic synthetic inputs and RNG-free weights shared by bench.py, smoke() and the GPU tests
(no dataset or checkpoint exists on the GPU box)."""
import math

import numpy as np
import torch


def synthetic_pcm(batch: int, length: int, seed: int = 1234) -> torch.Tensor:
    """pcm[b, n] = 0.25 sin(2 pi (200 + 37 (b mod 64)) n / 16000) + 0.05 u[b, n], u ~ U(-1, 1) (PCG64)."""
    rng = np.random.default_rng(seed)
    n = np.arange(length, dtype=np.float64)
    f = 200.0 + 37.0 * (np.arange(batch) % 64)
    tone = 0.25 * np.sin(2 * np.pi * f[:, None] * n[None, :] / 16000.0)
    noise = 0.05 * rng.uniform(-1.0, 1.0, size=(batch, length))
    return torch.from_numpy((tone + noise).astype(np.float32))


def closed_form(shape, scale, phase=1.0, freq=0.37) -> torch.Tensor:
    """w[k] = scale * sin(freq * k + phase): same formula as ``oracle.models.closed_form`` (kept separate so that the
    product never imports the oracle; ``tests/test_host.py`` checks the two agree)."""
    n = 1
    for s in shape:
        n *= s
    k = torch.arange(n, dtype=torch.float64)
    return (scale * torch.sin(freq * k + phase)).to(torch.float32).reshape(shape)


def res8_closed_form_state(num_labels: int, n_maps: int = 45):
    sd = {"conv0.weight": closed_form((n_maps, 1, 3, 3), 1.0 / 3.0, phase=0.3)}
    for i in range(1, 7):
        sd[f"conv{i}.weight"] = closed_form((n_maps, n_maps, 3, 3), math.sqrt(2.0 / (9 * n_maps)), phase=float(i))
    sd["output.weight"] = closed_form((num_labels, n_maps), 1.0 / math.sqrt(n_maps), phase=2.5)
    sd["output.bias"] = closed_form((num_labels,), 0.1, phase=0.7)
    return sd
