"""Headline benchmark: end-to-end training step of Howl's audio hot path on synthetic 16 kHz audio (BASELINE.json metric
"utterances/sec/node (res8, 1s@16kHz, 40-mel)").

    python bench.py --gpus N --steps K --warmup W [--config c3]

One step = one pass of the hot path over one batch already resident in HBM:
    PCM (B, L) -> fused log-mel frontend (+ZMUV) -> model forward -> loss -> backward
    -> [sum all-reduce of the flat gradient buffer over RCCL when N > 1] -> AdamW.

``--config`` selects the BASELINE.json configuration (default ``c3``, the one the metric is quoted on, at its per-GPU share):
    c1  res8 GSC-30,  64 x 1 s    (configs[0], the reference's own CPU-runnable case)
    c2  res8 hey-fire-fox (4 labels), 256 x 0.5 s   (configs[1])
    c3  res8 GSC-12, 512 x 1 s per GPU (4096 over 8 GPUs; configs[2])           <- the bench line
    c4  seq-lstm hey-fire-fox, CTC, 512 x 0.5 s      (configs[3])
    c5  mobilenet GSC-12, 512 x 1 s per GPU (2048 over 4 GPUs) with the timeshift/noise collate on the device (configs[4])
Weak scaling: the per-GPU batch is fixed as N grows.  Synthetic PCM and closed-form / seeded weights (no dataset or
checkpoint on the box).

Launch: under ``torch.distributed.run`` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) each process is one
rank; started plainly with ``--gpus N`` > 1 the script spawns the N ranks itself (one process per GPU, RCCL backend) and
relays rank 0's line.  ``HOWL_BENCH_BACKEND=gloo`` lets several ranks share one GPU (control-flow check; RCCL refuses that).

    eval  (f2, not a training step) batched streaming evaluation: FrameInferenceEngine.window_probabilities over 10 s clips,
          500 ms windows / 63 ms stride, against the reference's one-window-at-a-time loop (inference.py:223-267)

``--loop entry`` (res8 configurations) times the loop a user runs instead of the step on resident tensors: the epoch body of
``training.run.pretrain_gsc`` (``howl_amd.training.run.pretrain_gsc.train_epoch``) over a device-resident synthetic clip bank --
shuffled ids -> device collate (truncate, Timeshift, Noise, batchify; host draws in the reference's order) -> train-mode frontend
(VTLP draw) -> fused step -> loss to the workspace writer -- and reports it next to the resident-tensor step of the same process,
with the oracle of the same loop as ``cpu_baseline``.  ``--prewarm N`` (default 40): untimed steps in front of the W warm-up steps,
so that the contract's W + K bracket starts at the sustained clock.

Rank 0 prints ONE JSON line; besides the contract fields it carries
  "roofline":     the time-dominant kernel of the configuration: algorithmic FLOPs (or bytes) per launch / mean launch
                  duration measured with HIP events on the launch stream in a second pass of the same K steps
                  (res8: bwd_pair_kernel = data + weight gradient of a 45->45 layer in one launch, 58 % of the step, vs the
                  157.3 TFLOP/s fp32 MFMA peak, the forward convolution and the log-mel kernel beside it; seq-lstm:
                  recurrences + GEMMs vs the same peak; mobilenet: the fused convolution launches vs 8 TB/s HBM);
  "repeats":      the K timed steps as 5 consecutive segments (HIP events on the compute stream, no extra syncs):
                  median / min / max ms per step over the segments;
  "cpu_baseline": the oracle (CPU restatement of the reference step, torch-CPU) timed on this box's host cores on a
                  bounded sample: a thread sweep at batch 64 and a few steps at the bench's own batch (rank 0, N = 1 only);
  "eval_agreement": eval-mode logits of the trained bench model on a slice of the bench batch, HIP path vs the oracle at
                  identical weights (the "eval acc vs CPU ref" half of BASELINE.json's metric): argmax agreement and max
                  |logit difference| (rank 0, N = 1 only, with the CPU baseline leg);
  "like_for_like": (N = 1) the same K steps with the optimiser as its own launch, the structure every N > 1 step has.
  "rccl":         (N > 1) world size, backend, and the all-reduce of the flat gradient buffer timed on its own.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("NUM_MELS", "40")       # BASELINE.json's configurations; NUM_MELS=80 (stock Howl's default) runs too
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
MELS = int(os.environ["NUM_MELS"])

FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0

CONFIGS = {
    #      model       labels  batch/GPU  seconds  BASELINE.json entry
    "c1": ("res8", 30, 64, 1.0, "configs[0]: res8 GSC-30, batch 64, 1 s"),
    "c2": ("res8", 4, 256, 0.5, "configs[1]: res8 hey-fire-fox, batch 256, 0.5 s"),
    "c3": ("res8", 12, 512, 1.0, "configs[2] per-GPU share: res8 GSC-12, 512 x 1 s per GPU (4096 over 8)"),
    "c4": ("seq-lstm", 5, 512, 0.5, "configs[3]: seq-lstm hey-fire-fox CTC, batch 512, 0.5 s"),
    "c5": ("mobilenet", 12, 512, 1.0, "configs[4] per-GPU share: mobilenet GSC-12, 512 x 1 s per GPU (2048 over 4), device "
                                      "timeshift/noise collate in the step"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(CONFIGS) + ["eval"], default="c3")
    ap.add_argument("--batch-per-gpu", type=int, default=None, help="override the configuration's per-GPU batch")
    ap.add_argument("--seconds", type=float, default=None, help="override the utterance length")
    ap.add_argument("--labels", type=int, default=None)
    ap.add_argument("--prewarm", type=int, default=40, help="untimed steps in front of the W warmup steps (device clocks)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-unfused-leg", action="store_true",
                    help="skip the N = 1 `like_for_like` leg (the K steps once more with AdamW as its own launch)")
    ap.add_argument("--vtlp", action="store_true",
                    help="frontend in train mode, as the reference loop runs it (pretrain_gsc.py:120-126): per step one host draw, "
                         "VTLP-warped filterbank rebuilt on the device (howl_fb_from_points) on 75 %% of the steps")
    ap.add_argument("--loop", choices=["step", "entry"], default="step",
                    help="step (default, the bench line): the training step on a batch resident in HBM.  entry: the loop a user "
                         "runs -- training.run.pretrain_gsc's epoch body (howl_amd.training.run.pretrain_gsc.train_epoch: shuffled "
                         "clip ids -> device collate with Timeshift + Noise + batchify -> frontend in train mode (VTLP draw) -> fused "
                         "step -> loss logged through the workspace writer) over a device-resident synthetic clip bank, reported "
                         "beside the resident-tensor step of the same process (res8 configurations)")
    ap.add_argument("--no-lookahead", action="store_true",
                    help="seq-lstm: compute every batch's log-mel in its own step instead of inside the previous step's forward call")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="strong scaling: a FIXED global batch split over the N ranks (configs[2]: 4096); default is weak "
                         "scaling at the configuration's per-GPU batch")
    return ap.parse_args()


def c5_traffic():
    """HBM bytes per STEP of the MobileNet kernels from profiles/*c5_hbm_traffic*.txt (tools/pmc_round.sh c5: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, read side doubled per MI355X_MICROARCH.md)."""
    import re
    files = sorted((ROOT / "profiles").glob("*c5_hbm_traffic*.txt"))
    for f in reversed(files):
        m = re.search(r"total: read (\d+) MB \+ write (\d+) MB", f.read_text())
        if m:
            return (int(m.group(1)) + int(m.group(2))) * 1e6, f.name
    return None, None


# ---------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without a distributed environment starts the N ranks itself
# ---------------------------------------------------------------------------------------------------------------------
def spawn_ranks(n):
    import torch
    backend = os.environ.get("HOWL_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < n:
        raise SystemExit(f"bench.py --gpus {n}: only {have} HIP device(s) visible; one rank per GPU is required for RCCL")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HOWL_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    if any(codes):
        raise SystemExit(f"bench.py: rank exit codes {codes}")


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (oracle = checker code; only this leg and smoke() may touch it)
# ---------------------------------------------------------------------------------------------------------------------
def host_info():
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip() and phys is not None:
                cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return model, (len(cores) or os.cpu_count() or 1)


def cpu_baseline(model_name, L, C, budget_s, bench_batch=None):
    """The oracle's training step (frontend + forward + loss + backward + AdamW, mirrors pretrain_gsc.py:124-133 /
    train.py:286-302) on the host cores, on a bounded sample: batch 64 (configs[0]'s batch size), a few thread counts inside
    the time budget, the fastest reported together with the single-thread figure and the machine's physical core count."""
    import torch
    from oracle import frontend as ofe, models as om
    from howl_amd.utils.synth import synthetic_pcm
    B = 64 if model_name != "mobilenet" else 32
    pcm = synthetic_pcm(B, L)
    fb = ofe.mel_fb(MELS)
    z = ofe.Zmuv()
    z.update(ofe.standard_audio_transform(pcm[:2], fb))
    if model_name == "res8":
        labels = torch.arange(B) % C
        sd, names = om.res8_init(C), om.res8_param_names()
        opt = om.AdamWState([sd[n] for n in names], 0.01, 1e-5)

        def step():
            x = z(ofe.standard_audio_transform(pcm, fb))
            om.train_step(lambda s, xx: om.res8_forward(s, xx, True), sd, names, opt, x, labels)
    elif model_name == "seq-lstm":
        sd, names = om.lstm_init(C), om.lstm_param_names()
        opt = om.AdamWState([sd[n] for n in names], 1e-4, 1e-5)
        flen = ofe.compute_lengths(torch.full((B,), L))
        targets, tl = torch.tensor([[0, 1, 2]] * B), torch.tensor([3] * B)

        def step():
            x = z(ofe.standard_audio_transform(pcm, fb))
            params = [sd[n].requires_grad_(True) for n in names]
            scores, _ = om.seq_lstm_forward(sd, x, flen, aten=True)
            loss = torch.nn.functional.ctc_loss(torch.log_softmax(scores, -1), targets, flen, tl, blank=C - 1)
            grads = torch.autograd.grad(loss, params)
            for n in names:
                sd[n] = sd[n].detach()
            opt.step([sd[n] for n in names], grads)
    else:
        from oracle import mobilenet as omb
        labels = torch.arange(B) % C
        sd, names = omb.mobilenet_init(C), omb.mobilenet_param_names()
        opt = om.AdamWState([sd[n] for n in names], 0.001, 0.0)

        def step():
            x = z(ofe.standard_audio_transform(pcm, fb))
            om.train_step(lambda s, xx: omb.mobilenet_forward(s, xx, True), sd, names, opt, x, labels)

    cpu_model, physical = host_info()
    default_threads = torch.get_num_threads()
    # torch's default thread count oversubscribes these small convolutions on a many-core host: time a few counts inside
    # the budget and report the fastest with the count used; 1 thread gives the per-core figure
    candidates = sorted({1} | {min(default_threads, c) for c in (16, 32, 64)} | {min(default_threads, physical)})
    results = {}
    for nthreads in candidates:
        torch.set_num_threads(nthreads)
        for _ in range(2 if nthreads > 1 else 1):
            step()
        t0 = time.perf_counter()
        n = 0
        while n < 2 or (time.perf_counter() - t0 < budget_s / len(candidates) and n < 200):
            step()
            n += 1
        results[nthreads] = (B * n / (time.perf_counter() - t0), n)
    best = max(results, key=lambda k: results[k][0])
    # the same step at the bench's own batch ("same synthetic tensors, same step definition"), threads pinned
    # to the best count of the sweep: a few steps are enough (0.3-1 s each)
    at_bench = None
    if bench_batch and bench_batch != B and model_name == "res8":
        torch.set_num_threads(best)
        pcm_b = synthetic_pcm(bench_batch, L)
        labels_b = torch.arange(bench_batch) % C

        def step_b():
            x = z(ofe.standard_audio_transform(pcm_b, fb))
            om.train_step(lambda s, xx: om.res8_forward(s, xx, True), sd, names, opt, x, labels_b)

        step_b()
        t0, n = time.perf_counter(), 0
        while n < 3 or (time.perf_counter() - t0 < 4.0 and n < 10):
            step_b()
            n += 1
        at_bench = {"batch": bench_batch, "value": round(bench_batch * n / (time.perf_counter() - t0), 1), "threads": best,
                    "steps": n}
    torch.set_num_threads(default_threads)
    return {"value": round(results[best][0], 1), "unit": "utterances/sec", "cores": best, "kind": "port",
            "physical_cores": physical, "cpu_model": cpu_model, "value_1_thread": round(results[1][0], 1),
            "by_threads": {str(k): round(v[0], 1) for k, v in results.items()}, "at_bench_batch": at_bench,
            "sample": f"{results[best][1]} oracle training steps ({model_name}) of batch {B} x {L / 16000:g} s "
                      f"(torch-CPU; threads tried: {candidates}; box-to-box spread of this figure is large: 1.0-1.8 k utt/s "
                      f"on nominally identical hosts)"}


def eval_agreement(model_name, model, std, zmuv, pcm_dev, C, n=64):
    """The "eval acc vs CPU ref" half of the metric: eval-mode logits of the bench model (as trained by the timed steps) on the
    first n utterances of the bench batch, HIP path (frontend + model) vs the oracle (its own frontend + model) at identical
    weights and ZMUV statistics.  north_star: label indices bit-exact, logits within 1e-3."""
    import torch
    from oracle import frontend as ofe, models as om
    if model_name != "res8":
        return None
    n = min(n, pcm_dev.shape[0])
    was_training = model.training
    model.eval()
    with torch.no_grad():
        got = model(std.log_mel_for_model(pcm_dev[:n], zmuv), None).float().cpu()
    model.train(was_training)
    sd = {k: v.detach().float().cpu() if v.is_floating_point() else v.detach().cpu() for k, v in model.state_dict().items()}
    z = ofe.Zmuv()
    z.total, z.mean, z.mean2 = (t.detach().cpu() for t in (zmuv.total, zmuv.mean, zmuv.mean2))
    with torch.no_grad():
        ref = om.res8_forward(sd, z(ofe.standard_audio_transform(pcm_dev[:n].cpu(), ofe.mel_fb(MELS))), False)
    return {"utterances": n, "argmax_match": bool(torch.equal(got.argmax(1), ref.argmax(1))),
            "argmax_agreement": round((got.argmax(1) == ref.argmax(1)).float().mean().item(), 4),
            "max_abs_logit_diff": float(f"{(got - ref).abs().max().item():.3e}"), "tolerance": 1e-3,
            "weights": "the bench model after its timed training steps, eval-mode BatchNorm (running statistics)"}


def pmc_traffic(kernel="bwd_pair_kernel"):
    """HBM bytes per launch of a res8 kernel from the newest committed PMC summary (profiles/*pmc*.txt, written by
    tools/pmc_round.sh + tools/pmc_summary.py: separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE, in KB, read
    side doubled per the gfx950 correction in MI355X_MICROARCH.md).  Counters cannot be collected from inside this
    process, so the figure is the one measured on the same command line when the summary was taken (the file is named)."""
    import re
    files = sorted((ROOT / "profiles").glob("*pmc*.txt"), key=lambda f: ([int(x) for x in re.findall(r"\d+", f.name)], f.name))
    for f in reversed(files):        # the newest summary that lists this kernel (other summaries cover other configurations)
        vals = []
        lines = f.read_text().splitlines()
        for i, line in enumerate(lines):
            if line.startswith(kernel) and i + 1 < len(lines):
                d = json.loads(lines[i + 1].strip())
                if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
                    vals.append(((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0, d.get("launches", 1)))
        if vals:
            tot = sum(v * n for v, n in vals) / sum(n for _, n in vals)
            return tot, f.name
    return None, None


def cpu_entry_baseline(L, C, B, budget_s):
    """The oracle of the SAME loop (pretrain_gsc.py:120-133 with its collate, :78-80) on the host cores: timeshift + noise +
    batchify on CPU clips, train-mode frontend (VTLP filterbank on 75 % of the steps), res8 step.  Bounded sample."""
    import random
    import torch
    from oracle import collate as oc, frontend as ofe, models as om
    from howl_amd.utils.synth import synthetic_pcm
    Bc = min(B, 64)
    bank = synthetic_pcm(4 * Bc, L, seed=5)
    labels_all = [(i % 64) % C for i in range(4 * Bc)]
    rand, fb_std = random.Random(0), ofe.mel_fb(MELS)
    z = ofe.Zmuv()
    z.update(ofe.standard_audio_transform(bank[:2], fb_std))
    sd, names = om.res8_init(C), om.res8_param_names()
    opt = om.AdamWState([sd[n] for n in names], 0.01, 1e-5)
    cpu_model, physical = host_info()
    default_threads = torch.get_num_threads()
    torch.set_num_threads(min(default_threads, 16))
    perm = torch.randperm(4 * Bc, generator=torch.Generator().manual_seed(0)).tolist()

    def step(k):
        ids = perm[(k % 4) * Bc:(k % 4 + 1) * Bc]
        clips = oc.noise(rand, oc.timeshift(rand, oc.truncate_length([bank[i] for i in ids], L)))
        audio, lab, _, _ = oc.batchify(clips, [labels_all[i] for i in ids])
        fb = ofe.mel_fb(MELS, alpha=rand.random() * 0.2 + 0.9) if rand.random() < 0.75 else fb_std
        om.train_step(lambda s, xx: om.res8_forward(s, xx, True), sd, names, opt, z(ofe.standard_audio_transform(audio, fb)), lab)

    step(0)
    t0, n = time.perf_counter(), 0
    while n < 2 or (time.perf_counter() - t0 < budget_s and n < 200):
        step(n + 1)
        n += 1
    dt = time.perf_counter() - t0
    torch.set_num_threads(default_threads)
    return {"value": round(Bc * n / dt, 1), "unit": "utterances/sec", "cores": min(default_threads, 16), "kind": "port",
            "physical_cores": physical, "cpu_model": cpu_model,
            "sample": f"{n} iterations of the oracle's entry-point loop (collate chain + train-mode frontend + res8 step) at batch "
                      f"{Bc} x {L / 16000:g} s, torch-CPU, one process (the reference spreads the collate over cpu_count workers)"}


def bench_entry(args, dev, rank, world):
    """--loop entry: what `python -m training.run.pretrain_gsc` sustains per epoch, next to the resident-tensor step."""
    import random
    import tempfile
    import torch
    import torch.distributed as dist
    from howl_amd.data.collate import DeviceCollate
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    from howl_amd.model import RegisteredModel
    from howl_amd.training.data import ClipBank
    from howl_amd.training.fused import FusedTrainer
    from howl_amd.training.run.pretrain_gsc import train_epoch
    from howl_amd.utils.synth import res8_closed_form_state, synthetic_pcm
    from howl_amd.workspace import Workspace

    model_name, C, B, seconds, cfg_desc = CONFIGS[args.config]
    if model_name != "res8":
        raise SystemExit("bench.py --loop entry: res8 configurations (c1, c2, c3)")
    B = args.batch_per_gpu or B
    C = args.labels or C
    L = int(round((args.seconds or seconds) * 16000))
    n_batches = args.warmup + args.steps
    n_clips = max(4 * B, min(8192, B * n_batches))
    # the bank: tones + noise as everywhere, every tenth clip shorter than the window (as a GSC split has them)
    pcm = synthetic_pcm(n_clips, L, seed=4321 + rank)
    clips = [pcm[i][:L - 160 * (i % 37)] if i % 10 == 0 else pcm[i] for i in range(n_clips)]
    bank = ClipBank(clips, [(i % 64) % C for i in range(n_clips)], L, dev)
    del pcm, clips
    std = StandardAudioTransform().to(dev).eval()
    zmuv = ZmuvTransform().to(dev)
    zmuv.update(std(bank.audio[:8]))
    random.seed(1234)                 # VTLP's alpha draws (global `random`, transform.py:441)
    std.train()
    model = RegisteredModel.find_registered_class("res8")(C).to(dev)
    model.load_state_dict(res8_closed_form_state(C), strict=False)
    model.train()
    trainer = FusedTrainer(model, std, zmuv, lr=0.01, weight_decay=1e-5)
    trainer.broadcast_parameters()
    tmp = tempfile.mkdtemp(prefix="howl_bench_ws_")
    writer = Workspace(Path(tmp)).summary_writer if rank == 0 else Workspace(Path(tmp), writable=False).summary_writer

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def id_batches(n, seed):
        gen = torch.Generator().manual_seed(seed)
        out = []
        while len(out) < n:
            out += list(bank.index_batches(B, shuffle=True, drop_last=True, generator=gen))
        return out[:n]

    def timed(fn_warm, fn, reps=1):
        fn_warm()
        best = None
        for _ in range(reps):
            barrier()
            t0 = time.perf_counter()
            seen = fn()
            t_host = time.perf_counter() - t0          # everything enqueued
            barrier()
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, t_host, seen)
        return best

    results = {}
    for name, prefetch in (("prefetch", 2), ("inline", 0)):
        collate = DeviceCollate(bank.audio, bank.lengths_host, bank.labels, L, seed=99, replica=rank)
        run = lambda ids: train_epoch(trainer, collate, ids, std, writer, 0, prefetch=prefetch)
        results[name] = timed(lambda: run(id_batches(args.warmup, 1)), lambda: run(id_batches(args.steps, 2)))
    # the resident-tensor step of the bench line in the same process: same trainer, frontend in the same (train) mode
    pcm_res = bank.audio[:B].contiguous()
    labels_res = bank.labels[:B].contiguous()

    def resident(n):
        for _ in range(n):
            trainer.step(pcm_res, labels_res)
        return n * B

    results["resident"] = timed(lambda: resident(args.warmup), lambda: resident(args.steps))
    writer.close()
    mode = "prefetch" if results["prefetch"][0] <= results["inline"][0] else "inline"
    dt, t_host, seen = results[mode]
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    # host cost of one batch's collate preparation on its own (draws + sort + packed staging buffer)
    probe = DeviceCollate(bank.audio, bank.lengths_host, bank.labels, L, seed=7)
    ids = id_batches(1, 3)[0]
    t0 = time.perf_counter()
    for _ in range(50):
        probe.prepare(ids)
    prep_us = (time.perf_counter() - t0) / 50 * 1e6
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_entry_baseline(L, C, B, args.cpu_baseline_seconds)
    if rank == 0:
        ms = lambda r: round(r[0] / args.steps * 1e3, 4)
        res_ms = ms(results["resident"])
        out = {
            "metric": f"utterances/sec/node (res8 entry-point loop: device collate + train-mode frontend + training step, "
                      f"{L / 16000:g}s@16kHz, {MELS}-mel)",
            "value": round(seen * world / dt, 1), "unit": "utterances/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"training.run.pretrain_gsc's epoch body (howl_amd.training.run.pretrain_gsc.train_epoch) on a "
                                   f"device-resident bank of {n_clips} synthetic clips (every tenth shorter than the window): shuffled "
                                   f"ids -> DeviceCollate (truncate, Timeshift, Noise, batchify; host draws in the reference's order) -> "
                                   f"VTLP draw + log-mel -> res8 fwd + xent + bwd + AdamW -> loss to the workspace writer; "
                                   f"{B} utterances per GPU and step, {C} labels -- BASELINE {cfg_desc}",
                       "name": args.config, "loop": "entry", "collate": mode, "global_batch": B * world,
                       "samples_per_utterance": L, "labels": C, "parallelism": f"dp{world}" if world > 1 else "single",
                       "frontend": "vtlp-train"},
            "loop": {"ms_per_step_prefetch_thread": ms(results["prefetch"]), "ms_per_step_inline_collate": ms(results["inline"]),
                     "ms_per_step_resident_tensors": res_ms,
                     "entry_over_resident": round(res_ms / (dt / args.steps * 1e3), 4),
                     "host_enqueue_ms_per_step": round(t_host / args.steps * 1e3, 4),
                     "host_enqueue_ms_per_step_resident": round(results["resident"][1] / args.steps * 1e3, 4),
                     "collate_prepare_us_per_batch": round(prep_us, 1),
                     "note": "resident = trainer.step on one batch already in HBM (the default bench line's loop, frontend in train "
                             "mode here); host_enqueue = wall time until the last launch of the K steps was issued -- the entry loop's "
                             "staging ring lets the host run at most eight batches ahead, so there it ends 8 / K before the "
                             "device does whatever the host could do (host_enqueue_ms_per_step_resident is the host's own pace)"},
            "roofline": None, "cpu_baseline": cpu, "eval_agreement": None, "rccl": None,
        }
        print(json.dumps(out), flush=True)
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    if world > 1:
        dist.destroy_process_group()


def bench_eval(args, dev):
    """f2: the evaluation side of the metric.  The reference's ``FrameInferenceEngine`` scores a clip one 500 ms window per
    63 ms stride, each a batch-1 forward with a device->host copy (``inference.py:223-267``); the product scores all windows
    of a clip in ONE strided batch (``window_probabilities``).  Both paths run the same kernels here; the line reports
    windows/s of each over 10 s synthetic clips and their agreement."""
    import numpy as np
    import torch
    from howl_amd.context import InferenceContext
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    from howl_amd.model import RegisteredModel
    from howl_amd.model.inference import FrameInferenceEngine
    from howl_amd.utils import audio_utils
    from howl_amd.utils.synth import res8_closed_form_state, synthetic_pcm
    ctx = InferenceContext(["hey", "fire", "fox"], token_type="word", use_blank=False)
    model = RegisteredModel.find_registered_class("res8")(ctx.num_labels).to(dev)
    model.load_state_dict(res8_closed_form_state(ctx.num_labels), strict=False)
    model.eval()
    std = StandardAudioTransform().to(dev).eval()
    zmuv = ZmuvTransform().to(dev)
    clips = synthetic_pcm(8, 160000, seed=77).to(dev)                  # 8 clips of 10 s
    zmuv.update(std(clips[:1, :16000]))
    engine = FrameInferenceEngine(500, 63, model, zmuv, ctx)
    n_win = len(audio_utils.stride_starts(160000, 500, 63, 16000)[0])

    def batched():
        return [engine.window_probabilities(c) for c in clips]

    def many():        # the evaluation pass of training.run.train: all windows of all clips in one batch
        return engine.window_probabilities_many(list(clips))

    @torch.no_grad()
    def sequential():
        out = []
        for c in clips:
            rows = []
            for window in audio_utils.stride(c, 500, 63, 16000):
                if window.size(-1) < 1000:
                    break
                lengths = std.compute_lengths(torch.tensor([window.size(-1)]).to(dev))
                feats = std.log_mel_for_model(window.reshape(1, -1), zmuv)
                rows.append(model(feats, lengths).softmax(-1)[0].cpu().numpy())      # the per-window host copy of the reference loop
            out.append(np.stack(rows))
        return out

    res = {}
    for name, fn, reps in (("many", many, max(args.steps, 5)), ("batched", batched, max(args.steps, 5)), ("sequential", sequential, 2)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        res[name] = (len(clips) * n_win / dt, dt, out)
    diff = max(max(float(np.abs(a - b).max()) for a, b in zip(res[k][2], res["sequential"][2])) for k in ("many", "batched"))
    same = all(np.array_equal(a.argmax(1), b.argmax(1)) for k in ("many", "batched") for a, b in zip(res[k][2], res["sequential"][2]))
    print(json.dumps({
        "metric": "windows/sec (res8 streaming evaluation, 500 ms windows / 63 ms stride, 10 s clips)",
        "value": round(res["many"][0], 1), "unit": "windows/sec", "n_gpus": 1, "steps": max(args.steps, 5), "warmup": 1,
        "ms_per_step": round(res["many"][1] * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"FrameInferenceEngine.window_probabilities_many: {len(clips)} clips x 10 s, {n_win} windows each, all "
                               "clips in ONE batch (window gather + frontend + res8 eval forward + softmax + one host copy), as "
                               "training.run.train's evaluation pass scores a dataset", "name": "eval"},
        "one_batch_per_clip_windows_per_sec": round(res["batched"][0], 1),
        "sequential_windows_per_sec": round(res["sequential"][0], 1),
        "speedup_vs_one_window_per_launch": round(res["many"][0] / res["sequential"][0], 1),
        "agreement": {"argmax_match": same, "max_abs_prob_diff": float(f"{diff:.3e}")},
        "note": "sequential = the reference loop's structure (inference.py:223-267: per window a batch-1 forward and a "
                "device->host copy) on the same kernels"}), flush=True)


# ---------------------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)
        return
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    backend = os.environ.get("HOWL_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        assert torch.cuda.device_count() > local_rank, (f"rank {rank}: LOCAL_RANK={local_rank} but only "
                                                        f"{torch.cuda.device_count()} HIP device(s) visible")
    dev_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device(f"cuda:{dev_index}")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != world:      # fail loudly: a scaling number from fewer ranks than announced is worthless
            raise SystemExit(f"bench.py: the {backend} process group has {dist.get_world_size()} ranks, expected {world}")
    if args.config == "eval":
        return bench_eval(args, dev)
    if args.loop == "entry":
        return bench_entry(args, dev, rank, world)

    from howl_amd import lib as hlib
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    from howl_amd.model import RegisteredModel
    from howl_amd.training.fused import FusedTrainer
    from howl_amd.utils.synth import res8_closed_form_state, synthetic_pcm

    model_name, C, B, seconds, cfg_desc = CONFIGS[args.config]
    B = args.batch_per_gpu or B
    scaling = "weak"
    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit(f"bench.py: --global-batch {args.global_batch} is not a multiple of {world} ranks")
        B, scaling = args.global_batch // world, "strong"
    C = args.labels or C
    L = int(round((args.seconds or seconds) * 16000))
    pcm = synthetic_pcm(B, L, seed=1234 + rank).to(dev)
    labels = (torch.arange(B) % C).to(dev)

    std = StandardAudioTransform().to(dev).eval()   # eval-mode filterbank unless --vtlp
    zmuv = ZmuvTransform().to(dev)
    zmuv.update(std(pcm[:8]))
    if args.vtlp:
        import random
        random.seed(1234)       # the alpha draws (global `random`, transform.py:441): the same sequence on every rank
        std.train()
    torch.manual_seed(1234)     # seq-lstm / mobilenet start from torch's initialisers: the same weights (and final_loss) every run
    model = RegisteredModel.find_registered_class(model_name)(C).to(dev)
    if model_name == "res8":
        model.load_state_dict(res8_closed_form_state(C), strict=False)
    model.train()
    lr, wd = {"res8": (0.01, 1e-5), "seq-lstm": (1e-4, 1e-5), "mobilenet": (0.001, 0.0)}[model_name]   # envs/*.env
    trainer = FusedTrainer(model, std, zmuv, lr=lr, weight_decay=wd)
    trainer.broadcast_parameters()

    if model_name == "seq-lstm":
        n_frames = (L - 512) // 200 + 1                      # StandardAudioTransform.compute_lengths
        frame_lengths = torch.full((B,), n_frames).to(dev)   # the whole batch, lengths and targets included, is resident in HBM
        targets = torch.tensor([[0, 1, 2]] * B).to(dev)
        target_lengths = torch.tensor([3] * B).to(dev)

        # two resident batches, alternated: the look-ahead's "next batch" is a DIFFERENT tensor from the one the step trains on,
        # as a prefetching loader would hand it over (round 5 passed the same tensor twice)
        bufs = [pcm, synthetic_pcm(B, L, seed=4321 + rank).to(dev)]
        turn = [0]

        def step():    # frontend -> LSTM + head -> fused log_softmax + CTC(blank = C-1) -> backward -> flat AdamW
            # one-batch look-ahead (a prefetching loader's): the NEXT batch's frontend launch rides in this step's forward
            # recurrence (howl_lstm_fwd_next); every step still runs exactly one frontend pass and one model step
            cur, nxt = bufs[turn[0] & 1], bufs[(turn[0] + 1) & 1]
            turn[0] += 1
            return trainer.step_sequence(cur, frame_lengths, targets, target_lengths, C - 1, max_target=3, max_frames=n_frames,
                                         next_audio=None if args.no_lookahead else nxt)
    elif model_name == "mobilenet":
        from howl_amd.data.collate import DeviceCollate
        collate = DeviceCollate(pcm, torch.full((B,), L, dtype=torch.long), labels, max_len=L, seed=0, replica=rank)
        ids = list(range(B))

        def step():    # timeshift + white / salt-pepper noise + batchify on the device, then the training step
            batch = collate(ids)
            audio = batch.audio_data
            if audio.shape[1] != L:   # timeshift crops; the step geometry is fixed at L samples -> right-pad like batchify
                audio = torch.nn.functional.pad(audio, (0, L - audio.shape[1]))
            return trainer.step(audio, batch.labels)
    else:
        def step():
            return trainer.step(pcm, labels)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # clocks up before the contract's W warmup steps: the first ~30 ms of device work on an idle MI355X run below the sustained
    # clock (round 4: the first timed segment was 9 % slow with --warmup 5); untimed, outside the W + K bracket
    for _ in range(args.prewarm):
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    # the K timed steps are also cut into (up to) 5 consecutive segments by HIP events on the compute stream: no extra
    # synchronisation, the bracket below is the contract's
    n_seg = min(5, args.steps)
    bounds = [round(i * args.steps / n_seg) for i in range(n_seg + 1)]
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_seg + 1)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        if k in bounds:
            marks[bounds.index(k)].record()
        loss = step()
    marks[n_seg].record()
    barrier()
    dt = time.perf_counter() - t0
    seg_ms = sorted(marks[i].elapsed_time(marks[i + 1]) / (bounds[i + 1] - bounds[i]) for i in range(n_seg))
    repeats = {"segments": n_seg, "steps_per_segment": args.steps / n_seg, "ms_per_step_median": round(seg_ms[n_seg // 2], 4),
               "ms_per_step_min": round(seg_ms[0], 4), "ms_per_step_max": round(seg_ms[-1], 4),
               "timer": "HIP events on the compute stream between segments of the K timed steps (rank 0)"}
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    final_loss = loss.item()

    # Like-for-like denominator for a scaling curve: on ONE replica the optimiser step rides in the backward's last fold launch
    # (fused.py: world == 1), on N > 1 replicas it is a launch of its own behind the collective (1 / world scale + AdamW in one
    # kernel, queued behind the all-reduce without a host synchronisation).  At N = 1 the same K steps are therefore timed once
    # more with the optimiser as its own launch (HOWL_NO_FOLD_ADAMW: same bits, tests/test_gpu_lstm.py / test_gpu_res8.py).
    own_launch = None
    if world == 1 and model_name in ("res8", "seq-lstm") and not args.no_unfused_leg:
        os.environ["HOWL_NO_FOLD_ADAMW"] = "1"
        for _ in range(min(args.warmup, 3)):
            step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        own_launch = {"adamw_in_fold_at_n1": True,
                      "ms_per_step_adamw_own_launch": round((time.perf_counter() - t1) / args.steps * 1e3, 4),
                      "note": "N > 1 steps take the optimiser as its own launch behind the all-reduce: compare them with this line"}
        del os.environ["HOWL_NO_FOLD_ADAMW"]

    rccl = None
    if world > 1:
        # the step's one collective, timed on its own (HIP events on the compute stream, which the process group's stream
        # joins on both sides): the flat gradient buffer, sum over ranks
        buf = torch.zeros_like(trainer.fp.grad)
        for _ in range(5):
            dist.all_reduce(buf)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        e0.record()
        for _ in range(20):
            dist.all_reduce(buf)
        e1.record()
        torch.cuda.synchronize()
        # what the collectives cost the step: the same K steps once more with the gradient all-reduce left out (every rank
        # then steps on its local gradient: measurement only, the timed region above always reduces)
        trainer.skip_allreduce = True
        for _ in range(min(args.warmup, 3)):
            step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt_local = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(dt_local, op=dist.ReduceOp.MAX)
        trainer.skip_allreduce = False
        trainer.broadcast_parameters()          # the replicas diverged during the measurement: re-align before the roofline pass
        for t_ in (trainer.m, trainer.v):
            dist.broadcast(t_, 0)
        rccl = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                "allreduce_bytes": buf.numel() * 4, "allreduce_us": round(e0.elapsed_time(e1) * 1e3 / 20, 1),
                "collectives_per_step": trainer.collectives_last_step_reduced, "late_grads": trainer.late_grads,
                "ms_per_step_without_allreduce": round(dt_local.item() / args.steps * 1e3, 4),
                "allreduce_exposed_us": round((dt - dt_local.item()) / args.steps * 1e6, 1),
                "n1_step_has_adamw_in_fold": model_name in ("res8", "seq-lstm"),
                "note": "N = 1 folds AdamW into the backward's last launch; N > 1 runs it (with the 1 / world scale) as one launch "
                        "queued behind the all-reduce, no host sync in between: the N = 1 line's `like_for_like` block carries "
                        "the N = 1 time of THIS step structure"}

    roof = None
    lb = hlib.get()
    if not args.no_roofline:
        # second pass of the same K steps with HIP-event brackets around the dominant kernels.  EVERY rank runs it (the
        # step contains the gradient all-reduce: a rank stepping alone would wait for its peers forever); rank 0 reports.
        if rank == 0:
            lb.call("howl_profile_enable", 1)
        for _ in range(args.steps):
            step()
        barrier()
        if rank == 0:
            lb.call("howl_profile_enable", 0)
    if not args.no_roofline and rank == 0:

        def read(tag, reset=0):
            tot, cnt, work = ctypes.c_double(0), ctypes.c_int(0), ctypes.c_double(0)
            lb.call("howl_profile_read_work", tag.encode(), ctypes.byref(tot), ctypes.byref(cnt), ctypes.byref(work), reset)
            return tot.value, cnt.value, work.value

        T = 1 + L // 200
        tl, nl, _ = read("logmel")
        fe_bytes = (4.0 * L + 4.0 * MELS * T) * B
        logmel = {"avg_launch_ms": round(tl / max(nl, 1), 4),
                  "hbm_gbs": round(fe_bytes / (tl / max(nl, 1) * 1e-3) / 1e9, 1) if tl > 0 else None,
                  "hbm_frac": round(fe_bytes / (tl / max(nl, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if tl > 0 else None,
                  "algorithmic_bytes": round(fe_bytes)}
        if model_name == "res8":
            tf, nf, _ = read("conv3x3_fwd")
            tp, npair, _ = read("bwd_pair")
            t0f, n0f, _ = read("conv0_fwd")
            t0w, n0w, _ = read("conv0_wgrad")
            H = T // 3
            flops_launch = 2.0 * 9 * 45 * 45 * (H * (MELS // 4)) * B
            # the forward launches run alone on the device; dgrad and wgrad of a layer share ONE launch (half the CUs
            # each): its duration covers both and is listed under other_kernels
            fwd_ms = tf / max(nf, 1)
            fwd_tf = flops_launch / (fwd_ms * 1e-3) / 1e12 if fwd_ms > 0 else 0.0
            avg_ms = tp / max(npair, 1)
            achieved = 2 * flops_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
            full = B == 512 and L == 16000 and MELS == 40      # the configuration the committed PMC summary was taken on
            traffic, traffic_src = pmc_traffic("bwd_pair_kernel<1, 1, 0>") if full else (None, None)      # (the 40-bin, <= 83-frame instance)
            fwd_traffic, _ = pmc_traffic("conv3x3_mfma_kernel<0, 1, 0>") if full else (None, None)
            act = 4.0 * 45 * (H * (MELS // 4)) * B      # one (B, 45, H, M/4) fp32 map
            roof = {"bound": "mfma",
                    "kernel": "bwd_pair_kernel (data gradient + weight gradient of one 45->45 3x3 layer in ONE launch, half of "
                              "the CUs each, the BatchNorm / ReLU backward built inside both roles' tile staging): the "
                              "time-dominant kernel, 6 launches = 58 % of the step",
                    "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                    "traffic": None if traffic is None else round(traffic),
                    "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": traffic_src,
                    # per launch, averaged over the six layers: dx_i, s_i, s_{i-1} in (both roles stage them: the second
                    # reader hits L2), the skip gradient in and ds_i out on the three layers that have one, dx_{i-1} out,
                    # the weight-gradient partials out (one [48][416] row per weight-gradient workgroup) and the previous
                    # layer's partials in (folded by the data-gradient workgroups, five of the six launches)
                    "algorithmic_bytes": round(act * 5 + 4.0 * 48 * 416 * min(B, 128) * (1 + 5.0 / 6.0)),
                    "algorithmic_flops": round(2 * flops_launch),
                    "avg_launch_ms": round(avg_ms, 4), "launches": npair,
                    "other_kernels": {
                        "conv3x3_mfma_kernel<0> (45->45 3x3 convolution, forward launches; 6 = 30 % of the step)": {
                            "avg_launch_ms": round(fwd_ms, 4), "launches": nf, "tflops": round(fwd_tf, 2),
                            "frac": round(fwd_tf / FP32_MFMA_PEAK_TFLOPS, 4), "algorithmic_bytes": round(act * 2.5),
                            "traffic": None if fwd_traffic is None else round(fwd_traffic)},
                        "conv0 (1->45 3x3 + ReLU + AvgPool(3,4): forward / weight gradient launches)": {
                            "fwd_avg_launch_ms": round(t0f / max(n0f, 1), 4), "wgrad_avg_launch_ms": round(t0w / max(n0w, 1), 4),
                            "algorithmic_flops_each": round(2.0 * 9 * 45 * (3 * H) * MELS * B)},
                        "logmel": logmel}}
        elif model_name == "seq-lstm":
            parts = {t: read(t) for t in ("lstm_fwd", "lstm_bwd", "gemm")}
            tot_ms = sum(p[0] for p in parts.values())
            tot_fl = sum(p[2] for p in parts.values())
            per_step = args.steps
            achieved = tot_fl / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
            roof = {"bound": "mfma", "kernel": "lstm_fwd4/lstm_bwd4 recurrences + the GEMMs around them (input projection, "
                                               "head, weight gradients), summed",
                    "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                    "algorithmic_flops": round(tot_fl / per_step), "avg_launch_ms": round(tot_ms / per_step, 4),
                    "launches": sum(p[1] for p in parts.values()),
                    "note": "algorithmic FLOPs and summed kernel time per STEP (the recurrences are latency-bound: T dependent steps)",
                    "other_kernels": {k: {"ms_per_step": round(v[0] / per_step, 4), "launches_per_step": v[1] / per_step,
                                          "tflops": round(v[2] / (v[0] * 1e-3) / 1e12, 2) if v[0] > 0 else None}
                                      for k, v in parts.items()} | {"logmel": logmel}}
        else:
            ts, ns, wb = read("mb_conv")
            tg, ng, wf = read("gemm")
            achieved = wb / (ts * 1e-3) / 1e9 if ts > 0 else 0.0
            roof = {"bound": "hbm", "kernel": "the fused convolution launches of the 51 pointwise / depthwise layers (forward: "
                                              "conv + BatchNorm statistics, input normalised on load; backward: data + weight "
                                              "gradient + BatchNorm-backward reduction in one launch), summed",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": None if c5_traffic()[0] is None or (B, L) != (512, 16000) else round(c5_traffic()[0]),
                    "traffic_unit": "HBM bytes per STEP, all MobileNet kernels (PMC FETCH_SIZE x2 + WRITE_SIZE)",
                    "traffic_source": c5_traffic()[1],
                    "algorithmic_bytes": round(wb / args.steps), "avg_launch_ms": round(ts / max(ns, 1), 4), "launches": ns,
                    "note": "algorithmic bytes (every operand read once, every result written once) and summed kernel time per "
                            "STEP; most layers are a few MB and latency-bound, see DESIGN.md 5c",
                    "other_kernels": {"gemm (classifier)": {"ms_per_step": round(tg / args.steps, 4),
                                                            "launches_per_step": ng / args.steps},
                                      "mb_conv_ms_per_step": round(ts / args.steps, 4),
                                      "mb_conv_launches_per_step": ns / args.steps, "logmel": logmel}}
        read("logmel", reset=1)

    cpu = agree = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        agree = eval_agreement(model_name, model, std, zmuv, pcm, C)
        cpu = cpu_baseline(model_name, L, C, args.cpu_baseline_seconds, bench_batch=B)

    if rank == 0:
        total_utts = B * world * args.steps
        step_desc = {"res8": "frontend+fwd+xent+bwd+AdamW", "seq-lstm": "frontend+LSTM+head+CTC+bwd+AdamW",
                     "mobilenet": "device collate (timeshift+noise)+frontend+fwd+xent+bwd+AdamW"}[model_name]
        out = {
            "metric": f"utterances/sec/node ({model_name} end-to-end training step, {L / 16000:g}s@16kHz, {MELS}-mel)",
            "value": round(total_utts / dt, 1), "unit": "utterances/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{model_name} training step ({step_desc}), {B} x {L / 16000:g} s utterances per GPU, "
                                   f"{C} labels -- BASELINE {cfg_desc}"
                                   + ("; frontend in train mode (VTLP filterbank on 75 % of the steps)" if args.vtlp else "")
                                   + ("; one-batch look-ahead: the next batch's log-mel runs as rider blocks of this step's forward "
                                      "recurrence launch" if model_name == "seq-lstm" and not args.no_lookahead and not args.vtlp else "")
                                   + ("; parity unpinned (torchvision absent: oracle restates the published architecture)"
                                      if model_name == "mobilenet" else ""),
                       "name": args.config, "global_batch": B * world, "samples_per_utterance": L, "labels": C,
                       "parallelism": f"dp{world}" if world > 1 else "single", "frontend": "vtlp-train" if args.vtlp else "eval"},
            "final_loss": round(final_loss, 5), "repeats": repeats,
            "roofline": roof, "cpu_baseline": cpu, "eval_agreement": agree, "rccl": rccl,
            "like_for_like": own_launch,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
