"""Headline benchmark: end-to-end res8 training step on synthetic 16 kHz audio (BASELINE.json metric
"utterances/sec/node (res8, 1s@16kHz, 40-mel)").

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch already resident in HBM:
    PCM (B,16000) -> fused log-mel frontend (+ZMUV) -> res8 forward (training-mode BN) -> cross-entropy
    -> res8 backward -> [sum all-reduce of the flat 441 KB gradient over RCCL when N > 1] -> AdamW.
Workload per GPU: BASELINE.json configs[2] at its per-GPU share -- res8, 12 labels, 512 utterances of 1 s per GPU
(global batch 4096 at 8 GPUs); weak scaling.  Synthetic PCM and closed-form weights (no dataset / checkpoint on the box).

Rank 0 prints ONE JSON line; it also carries
  "roofline":     the dominant kernel (MFMA conv3x3 45->45, forward+dgrad launches) -- algorithmic FLOPs per launch
                  (2*9*45*45*270 per utterance x B) / mean launch duration measured with HIP events on the launch stream,
                  against the 157.3 TFLOP/s fp32 MFMA peak;
  "cpu_baseline": the oracle (CPU restatement of the reference step, torch-CPU) timed on this box's host cores on a
                  bounded sample (rank 0, N = 1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("NUM_MELS", "40")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch-per-gpu", type=int, default=512)
    ap.add_argument("--seconds", type=float, default=1.0, help="utterance length")
    ap.add_argument("--labels", type=int, default=12)
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def cpu_baseline(L, C, budget_s):
    """Oracle training step (frontend + res8 fwd/bwd + AdamW, mirrors pretrain_gsc.py:124-133) on the host cores."""
    from oracle import frontend as ofe, models as om
    from howl_amd.utils.synth import synthetic_pcm
    B = 64  # BASELINE.json configs[0] batch size
    pcm = synthetic_pcm(B, L)
    labels = torch.arange(B) % C
    fb = ofe.mel_fb(40)
    z = ofe.Zmuv()
    z.update(ofe.standard_audio_transform(pcm[:2], fb))
    sd = om.res8_init(C)
    names = om.res8_param_names()
    opt = om.AdamWState([sd[n] for n in names], 0.01, 1e-5)

    def step():
        x = z(ofe.standard_audio_transform(pcm, fb))
        om.train_step(lambda s, xx: om.res8_forward(s, xx, True), sd, names, opt, x, labels)

    # torch's default thread count (= physical cores) oversubscribes these small convolutions on a many-core host:
    # time a few thread counts inside the budget and report the fastest, with the count used
    default_threads = torch.get_num_threads()
    candidates = sorted({min(default_threads, c) for c in (16, 32, 64)} | {default_threads})
    best = None
    for nthreads in candidates:
        torch.set_num_threads(nthreads)
        for _ in range(2):
            step()
        t0 = time.perf_counter()
        n = 0
        while n < 3 or (time.perf_counter() - t0 < budget_s / len(candidates) and n < 200):
            step()
            n += 1
        dt = time.perf_counter() - t0
        rate = B * n / dt
        if best is None or rate > best[0]:
            best = (rate, nthreads, n)
    torch.set_num_threads(default_threads)
    rate, nthreads, n = best
    return {"value": round(rate, 1), "unit": "utterances/sec", "cores": nthreads, "kind": "port",
            "sample": f"{n} oracle training steps of batch {B} x {L / 16000:g} s (torch-CPU, best of {candidates} threads)"}


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary (profiles/*pmc*.txt, written by
    tools/pmc_round.sh + tools/pmc_summary.py: separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE, in KB, read
    side doubled per the gfx950 correction in MI355X_MICROARCH.md).  Counters cannot be collected from inside this
    process, so the figure is the one measured on the same command line when the summary was taken."""
    import re
    files = sorted((ROOT / "profiles").glob("*pmc*.txt"), key=lambda f: [int(x) for x in re.findall(r"\d+", f.name)])
    if not files:
        return None, None
    vals = []
    lines = files[-1].read_text().splitlines()
    for i, line in enumerate(lines):
        if line.startswith("conv3x3_mfma_kernel<0>") and i + 1 < len(lines):
            d = json.loads(lines[i + 1].strip())
            if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
                vals.append(((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0, d.get("launches", 1)))
    if not vals:
        return None, None
    tot = sum(v * n for v, n in vals) / sum(n for _, n in vals)
    return tot, files[-1].name


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    # HOWL_BENCH_BACKEND=gloo lets the multi-rank control flow be exercised with several ranks on ONE GPU (RCCL refuses two
    # ranks per device); the default, and what the driver runs, is one rank per GPU over RCCL
    backend = os.environ.get("HOWL_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{dev_index}"))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(dev_index)
    dev = torch.device(f"cuda:{dev_index}")

    from howl_amd import lib as hlib
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    from howl_amd.model import RegisteredModel
    from howl_amd.training.fused import FusedRes8Trainer
    from howl_amd.utils.synth import res8_closed_form_state, synthetic_pcm

    B, C = args.batch_per_gpu, args.labels
    L = int(round(args.seconds * 16000))
    pcm = synthetic_pcm(B, L, seed=1234 + rank).to(dev)
    labels = (torch.arange(B) % C).to(dev)

    std = StandardAudioTransform().to(dev).eval()   # eval-mode filterbank; VTLP is exercised by the tests
    zmuv = ZmuvTransform().to(dev)
    zmuv.update(std(pcm[:8]))
    model = RegisteredModel.find_registered_class("res8")(C).to(dev)
    model.load_state_dict(res8_closed_form_state(C), strict=False)
    model.train()
    trainer = FusedRes8Trainer(model, std, zmuv, lr=0.01, weight_decay=1e-5)
    trainer.broadcast_parameters()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(pcm, labels)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(pcm, labels)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    final_loss = loss.item()

    roof = None
    if not args.no_roofline:
        # second pass of the same K steps with HIP-event brackets around the dominant kernels.  EVERY rank runs it (the
        # step contains the gradient all-reduce: a rank stepping alone would wait for its peers forever); rank 0 reports.
        lb = hlib.get()
        if rank == 0:
            lb.call("howl_profile_enable", 1)
        for _ in range(args.steps):
            trainer.step(pcm, labels)
        barrier()
        if rank == 0:
            lb.call("howl_profile_enable", 0)
    if not args.no_roofline and rank == 0:

        def read(tag, reset=0):
            tot, cnt = ctypes.c_double(0), ctypes.c_int(0)
            lb.call("howl_profile_read", tag.encode(), ctypes.byref(tot), ctypes.byref(cnt), reset)
            return tot.value, cnt.value

        tf, nf = read("conv3x3_fwd")
        td, nd = read("conv3x3_dgrad")
        tw, nw = read("wgrad")
        tl, nl = read("logmel", reset=1)
        H = (1 + L // 200) // 3
        flops_launch = 2.0 * 9 * 45 * 45 * (H * 10) * B
        # the forward launches run alone on the device; dgrad and wgrad of a layer share it (two HIP queues, half the CUs
        # each), so their event-bracketed durations include the sharing and are listed under other_kernels only
        avg_ms = tf / max(nf, 1)
        achieved = flops_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic() if (B == 512 and L == 16000) else (None, None)
        roof = {"bound": "mfma", "kernel": "conv3x3_mfma_kernel<0> (45->45 3x3 convolution, forward launches)",
                "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                "traffic": None if traffic is None else round(traffic),
                "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": traffic_src,
                # input + output maps always; the residual on half of the forward launches
                "algorithmic_bytes": round(4.0 * 45 * (H * 10) * B * 2.5),
                "avg_launch_ms": round(avg_ms, 4), "launches": nf,
                "other_kernels": {
                    "note": "dgrad and wgrad of a layer run concurrently on half the CUs each: per-launch durations overlap",
                    "conv3x3_dgrad": {"avg_launch_ms": round(td / max(nd, 1), 4), "launches": nd},
                    "wgrad_mfma": {"avg_launch_ms": round(tw / max(nw, 1), 4), "launches": nw},
                    "dgrad+wgrad_pair": {"tflops": round(2 * flops_launch / (max(td / max(nd, 1), tw / max(nw, 1)) * 1e-3) / 1e12, 2)
                                         if tw > 0 and td > 0 else None},
                    "logmel": {"avg_launch_ms": round(tl / max(nl, 1), 4),
                               "hbm_gbs": round((4.0 * L + 4.0 * 40 * (1 + L // 200)) * B / (tl / max(nl, 1) * 1e-3) / 1e9, 1)
                               if tl > 0 else None, "hbm_frac": round((4.0 * L + 4.0 * 40 * (1 + L // 200)) * B /
                                                                      (tl / max(nl, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                               if tl > 0 else None}}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(L, C, args.cpu_baseline_seconds)

    if rank == 0:
        total_utts = B * world * args.steps
        out = {
            "metric": "utterances/sec/node (res8 end-to-end training step, 1s@16kHz, 40-mel)",
            "value": round(total_utts / dt, 1), "unit": "utterances/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"res8 GSC-12 training step (frontend+fwd+loss+bwd+AdamW), {B} x {L / 16000:g} s "
                                   f"utterances per GPU (BASELINE configs[2] per-GPU share)",
                       "global_batch": B * world, "samples_per_utterance": L, "labels": C,
                       "parallelism": f"dp{world}" if world > 1 else "single"},
            "final_loss": round(final_loss, 5),
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
