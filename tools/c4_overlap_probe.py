"""Round 5 probe (tools only): do the 128 CUs the four-sequence LSTM recurrence leaves idle at B = 512 take other work at full
speed?  Queues N forward recurrences on one stream and 3N head-forward GEMMs (19,456 x 128 -> 256 -> 5) on another; prints the
wall time of each alone and of both together."""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("NUM_MELS", "40")
import torch
from howl_amd.model import rnn

dev = "cuda"
B, T, M, N = 512, 38, 40, 30
x = torch.randn(B, T, M, device=dev)
lengths = torch.full((B,), T, dtype=torch.int64, device=dev)
w_ih, w_hh, b = torch.randn(512, 40, device=dev) * 0.1, torch.randn(512, 128, device=dev) * 0.1, torch.zeros(512, device=dev)
hs = torch.randn(B, T, 128, device=dev)
w1, b1, w2, b2 = torch.randn(256, 128, device=dev) * 0.1, torch.zeros(256, device=dev), torch.randn(5, 256, device=dev) * 0.1, torch.zeros(5, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def rec(n):
    with torch.cuda.stream(s1):
        for _ in range(n):
            rnn._lstm_forward_raw(x, lengths, T, None, None, w_ih, w_hh, b, b)


def head(n):
    with torch.cuda.stream(s2):
        for _ in range(n):
            rnn._head_forward_raw(hs, w1, b1, w2, b2)


def wall(fn):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e6


t_rec = wall(lambda: rec(N))
t_head = wall(lambda: head(3 * N))
t_both = wall(lambda: (rec(N), head(3 * N)))
t_both2 = wall(lambda: (head(3 * N), rec(N)))
print(f"recurrence x{N}: {t_rec:.0f} us ({t_rec / N:.1f} each); head fwd x{3 * N}: {t_head:.0f} us ({t_head / 3 / N:.1f} each); "
      f"both streams: {t_both:.0f} us (rec queued first) / {t_both2:.0f} us (head queued first); sum {t_rec + t_head:.0f}, max {max(t_rec, t_head):.0f}")
