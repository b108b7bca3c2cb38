"""Shared cases for the CTC kernel tests: random logits with ragged input lengths, repeated labels, an empty target,
a target that only just fits (T_b = L + repeats) and one that cannot be aligned (loss = inf, excluded from the mean check).
The reference is the reference's own arithmetic: torch's log_softmax + ctc_loss on the CPU (train.py:250-256,291-296)."""
import numpy as np
import torch


def make_case(T, B, C, Lmax, seed, blank=None, tight=False):
    rng = np.random.default_rng(seed)
    blank = C - 1 if blank is None else blank
    logits = torch.from_numpy(rng.normal(0, 2.0, (B, T, C)).astype(np.float32))
    in_len = np.sort(rng.integers(max(1, T // 2), T + 1, B))[::-1].copy()
    in_len[0] = T
    tgt_len = rng.integers(0, Lmax + 1, B)
    labels = [c for c in range(C) if c != blank]
    targets = np.zeros((B, max(Lmax, 1)), np.int64)
    for b in range(B):
        row = rng.choice(labels, tgt_len[b])
        if tgt_len[b] >= 2 and b % 2 == 0:
            row[1] = row[0]                                   # a repeated label: needs the blank in between
        need = tgt_len[b] + int(np.sum(row[1:] == row[:-1]))  # shortest input that admits an alignment
        if in_len[b] < need:
            in_len[b] = min(T, need)
        if tight and b == 1:
            in_len[b] = max(1, min(T, need))
        targets[b, : tgt_len[b]] = row
    return logits, torch.from_numpy(targets), torch.from_numpy(in_len.astype(np.int64)), torch.from_numpy(tgt_len.astype(np.int64)), blank


def reference(logits_btc, targets, in_len, tgt_len, blank):
    z = logits_btc.clone().requires_grad_(True)
    lp = torch.log_softmax(z.permute(1, 0, 2), -1)
    per = torch.nn.functional.ctc_loss(lp, targets, in_len, tgt_len, blank, reduction="none")
    loss = torch.nn.functional.ctc_loss(lp, targets, in_len, tgt_len, blank)
    loss.backward()
    return per.detach(), loss.detach(), z.grad.detach()
