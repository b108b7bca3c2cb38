"""Oracle: res8 / lstm / seq-lstm classifiers, loss and optimiser step.  Test infrastructure only.

Functional restatement (torch-CPU, autograd for the backward) of
``howl/model/cnn.py:107-145`` (Res8), ``howl/model/rnn.py:41-91`` (SequentialLstm, SimpleLstm) and
of the step in ``training/run/pretrain_gsc.py:124-133`` (CrossEntropyLoss + AdamW).
Parameters are passed as a ``state_dict``-shaped mapping with the reference's key names
(res8: ``conv0.weight``, ``conv{i}.weight``, ``bn{i}.running_mean|running_var|num_batches_tracked``,
``output.weight|bias``; lstm: ``lstm.weight_ih_l0 ...``, ``dnn.0.*``, ``dnn.2.*``).
"""
import math
from typing import Dict

import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

N_MAPS = 45
N_LAYERS = 6
POOLING = (3, 4)
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
HIDDEN = 128


def closed_form(shape, scale, phase=1.0, freq=0.37):
    """Deterministic RNG-free weights: w[k] = scale * sin(freq*k + phase)."""
    n = 1
    for s in shape:
        n *= s
    k = torch.arange(n, dtype=torch.float64)
    return (scale * torch.sin(freq * k + phase)).to(torch.float32).reshape(shape)


def res8_init(num_labels: int) -> Dict[str, torch.Tensor]:
    """Closed-form res8 parameters + fresh BN buffers, keyed like the reference ``state_dict``."""
    sd = {"conv0.weight": closed_form((N_MAPS, 1, 3, 3), 1.0 / 3.0, phase=0.3)}
    for i in range(1, N_LAYERS + 1):
        sd[f"bn{i}.running_mean"] = torch.zeros(N_MAPS)
        sd[f"bn{i}.running_var"] = torch.ones(N_MAPS)
        sd[f"bn{i}.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
        sd[f"conv{i}.weight"] = closed_form((N_MAPS, N_MAPS, 3, 3), math.sqrt(2.0 / (9 * N_MAPS)), phase=float(i))
    sd["output.weight"] = closed_form((num_labels, N_MAPS), 1.0 / math.sqrt(N_MAPS), phase=2.5)
    sd["output.bias"] = closed_form((num_labels,), 0.1, phase=0.7)
    return sd


def res8_param_names():
    return ["conv0.weight"] + [f"conv{i}.weight" for i in range(1, N_LAYERS + 1)] + ["output.weight", "output.bias"]


def res8_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, training: bool, pre_relu=None, relu_masks=None) -> torch.Tensor:
    """``Res8.forward`` (``cnn.py:127-145``).  x: (B, C>=1, M, T).  In training mode the BN buffers in
    ``sd`` are updated in place exactly like ``nn.BatchNorm2d(affine=False)`` (momentum 0.1,
    unbiased variance for the running estimate, ``num_batches_tracked += 1``).
    Test diagnostics (a ReLU whose input lies within fp32 rounding of zero comes out on either side depending on the summation
    order, and one such flip moves small-batch gradients by O(1 / positions) -- for this function against its own fp64 run as much
    as for a kernel): ``pre_relu`` (a list) receives every convolution's output z, (B, 45, T, M) for conv0 and (B, 45, T/3, M/4) for
    the rest; ``relu_masks`` (seven bool tensors of those shapes) replaces relu(z) by z * mask, i.e. the same function with the
    on/off decisions taken from elsewhere."""
    x = x[:, :1]
    x = x.permute(0, 1, 3, 2).contiguous()
    old_x = None
    for i in range(N_LAYERS + 1):
        z = F.conv2d(x, sd[f"conv{i}.weight"], None, padding=1)
        if pre_relu is not None:
            pre_relu.append(z.detach())
        y = F.relu(z) if relu_masks is None else z * relu_masks[i].to(z.dtype)
        if i == 0:
            y = F.avg_pool2d(y, POOLING)
            old_x = y
        if i > 0 and i % 2 == 0:
            x = y + old_x
            old_x = x
        else:
            x = y
        if i > 0:
            if training:
                sd[f"bn{i}.num_batches_tracked"] += 1
            x = F.batch_norm(x, sd[f"bn{i}.running_mean"], sd[f"bn{i}.running_var"], None, None,
                             training, BN_MOMENTUM, BN_EPS)
    x = x.view(x.size(0), x.size(1), -1)
    x = torch.mean(x, 2)
    return F.linear(x, sd["output.weight"], sd["output.bias"])


def lstm_init(num_labels: int, num_mels: int = 40) -> Dict[str, torch.Tensor]:
    k = 1.0 / math.sqrt(HIDDEN)
    return {
        "lstm.weight_ih_l0": closed_form((4 * HIDDEN, num_mels), k, phase=0.1),
        "lstm.weight_hh_l0": closed_form((4 * HIDDEN, HIDDEN), k, phase=0.2),
        "lstm.bias_ih_l0": closed_form((4 * HIDDEN,), k, phase=0.3),
        "lstm.bias_hh_l0": closed_form((4 * HIDDEN,), k, phase=0.4),
        "dnn.0.weight": closed_form((2 * HIDDEN, HIDDEN), k, phase=0.5),
        "dnn.0.bias": closed_form((2 * HIDDEN,), k, phase=0.6),
        "dnn.2.weight": closed_form((num_labels, 2 * HIDDEN), 1.0 / math.sqrt(2 * HIDDEN), phase=0.7),
        "dnn.2.bias": closed_form((num_labels,), 0.05, phase=0.8),
    }


def lstm_param_names():
    return ["lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0",
            "dnn.0.weight", "dnn.0.bias", "dnn.2.weight", "dnn.2.bias"]


def _lstm_cell_seq(sd, x, lengths, hx):
    """Packed-sequence LSTM semantics written out: gate order i, f, g, o; a sequence stops updating
    (h, c) after its own length; padded outputs are zero (``pad_packed_sequence``)."""
    T, B, _ = x.shape
    w_ih, w_hh = sd["lstm.weight_ih_l0"], sd["lstm.weight_hh_l0"]
    b = sd["lstm.bias_ih_l0"] + sd["lstm.bias_hh_l0"]
    h = x.new_zeros(B, HIDDEN) if hx is None else hx[0][0]
    c = x.new_zeros(B, HIDDEN) if hx is None else hx[1][0]
    T_out = T if lengths is None else int(lengths.max())
    outs = []
    for t in range(T_out):
        g = x[t] @ w_ih.t() + h @ w_hh.t() + b
        i, f, gg, o = g.chunk(4, 1)
        c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h_new = torch.sigmoid(o) * torch.tanh(c_new)
        if lengths is not None:
            m = (lengths > t).to(x.dtype).unsqueeze(1)
            c = m * c_new + (1 - m) * c
            h = m * h_new + (1 - m) * h
            outs.append(m * h_new)
        else:
            c, h = c_new, h_new
            outs.append(h_new)
    return torch.stack(outs), (h.unsqueeze(0), c.unsqueeze(0))


def _lstm_aten(sd, x, lengths, hx):
    """Same thing through ATen exactly as the reference calls it (``rnn.py:63-70,88``)."""
    lstm = torch.nn.LSTM(x.size(-1), HIDDEN)
    with torch.no_grad():
        for k in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
            getattr(lstm, k).copy_(sd["lstm." + k])
    inp = pack_padded_sequence(x, lengths) if lengths is not None else x
    seq, hc = torch.func.functional_call(lstm, {k: sd["lstm." + k] for k in
                                                ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")},
                                         (inp, hx))
    if lengths is not None:
        seq, _ = pad_packed_sequence(seq)
    return seq, hc


def _dnn(sd, x):
    return F.linear(F.relu(F.linear(x, sd["dnn.0.weight"], sd["dnn.0.bias"])), sd["dnn.2.weight"], sd["dnn.2.bias"])


def seq_lstm_forward(sd, x, lengths, hx=None, aten=False):
    """``SequentialLstm.forward`` (``rnn.py:60-71``): (B, C, M, T) -> (T_len, B, num_labels); also returns (h, c)."""
    x = x[:, 0].permute(2, 0, 1).contiguous()
    seq, hc = (_lstm_aten if aten else _lstm_cell_seq)(sd, x, lengths, hx)
    return _dnn(sd, seq), hc


def lstm_forward(sd, x, lengths, hx=None, aten=False):
    """``SimpleLstm.forward`` (``rnn.py:85-91``): head on the final hidden state -> (B, num_labels)."""
    x = x[:, 0].permute(2, 0, 1).contiguous()
    _, hc = (_lstm_aten if aten else _lstm_cell_seq)(sd, x, lengths, hx)
    return _dnn(sd, hc[0].squeeze(0)), hc


class AdamWState:
    """torch.optim.AdamW defaults as used at ``pretrain_gsc.py:93`` / ``train.py:256``
    (betas (0.9, 0.999), eps 1e-8, decoupled weight decay), written out per tensor."""

    def __init__(self, params, lr, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8):
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]
        self.t = 0

    def step(self, params, grads):
        self.t += 1
        b1, b2 = self.betas
        bc1 = 1 - b1 ** self.t
        bc2 = 1 - b2 ** self.t
        with torch.no_grad():
            for p, g, m, v in zip(params, grads, self.m, self.v):
                p.mul_(1 - self.lr * self.wd)
                m.mul_(b1).add_(g, alpha=1 - b1)
                v.mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
                p.addcdiv_(m, denom, value=-self.lr / bc1)


def train_step(forward_fn, sd, names, opt: AdamWState, x, labels):
    """One ``pretrain_gsc.py:126-133`` step on already-extracted features: forward (training mode),
    mean cross-entropy, backward, AdamW.  Returns (loss, logits, grads-by-name)."""
    params = [sd[n].requires_grad_(True) for n in names]
    logits = forward_fn(sd, x)
    loss = F.cross_entropy(logits, labels)
    grads = torch.autograd.grad(loss, params)
    for n in names:
        sd[n] = sd[n].detach()
    opt.step([sd[n] for n in names], grads)
    return loss.detach(), logits.detach(), dict(zip(names, grads))
