"""world_size-2 gloo test (CPU) of the data-parallel path: shard by utterance, sum all-reduce of the flat gradient,
mean applied by the optimiser, replicas stay bit-identical.  The per-rank forward/backward is the oracle (the HIP
kernels need a GPU); what is under test is howl_amd.parallel + the flat-buffer protocol the fused trainer uses."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NUM_MELS="40")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from howl_amd import parallel
    from oracle import models as om
    B, C = 8, 4
    torch.manual_seed(0)
    x = torch.randn(B, 3, 40, 41)
    labels = torch.arange(B) % C
    lo, hi = parallel.shard_range(B, rank, world)
    sd = om.res8_init(C)
    names = om.res8_param_names()
    params = [sd[n].clone().requires_grad_(True) for n in names]
    sdl = dict(sd)
    sdl.update(dict(zip(names, params)))
    loss = torch.nn.functional.cross_entropy(om.res8_forward(sdl, x[lo:hi], True), labels[lo:hi])
    grads = torch.autograd.grad(loss, params)
    flat = torch.cat([g.reshape(-1) for g in grads])
    local = flat.clone()
    scale = parallel.allreduce_sum_(flat)
    # every rank applies the same averaged gradient with the same optimiser state -> identical weights
    flat_p = torch.cat([p.detach().reshape(-1) for p in params])
    opt = om.AdamWState([flat_p], 0.01, 1e-5)
    opt.step([flat_p], [flat * scale])
    gathered = [torch.zeros_like(flat_p) for _ in range(world)]
    dist.all_gather(gathered, flat_p)
    locals_ = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(locals_, local)
    if rank == 0:
        out["scale"] = scale
        out["identical"] = all(torch.equal(gathered[0], g) for g in gathered)
        out["mean_err"] = (flat * scale - sum(locals_) / world).abs().max().item()
        out["range"] = (lo, hi)
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce():
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        assert out["scale"] == 0.5 and out["identical"] and out["mean_err"] < 1e-7 and out["range"] == (0, 4)


def test_shard_range_covers_batch():
    from howl_amd.parallel import shard_range
    for B in (1, 7, 512, 4096):
        for world in (1, 2, 3, 8):
            spans = [shard_range(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _trainer_worker(rank, world, port, out, late_grads="overlap"):
    """The product's FusedTrainer (flat parameter / gradient buffers, broadcast, in-step all-reduce, fused AdamW) with its
    kernels running on the hipemu build, one process per rank over gloo."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NUM_MELS="40")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from emu_util import emulated_package
    from howl_amd import parallel
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    from howl_amd.model import RegisteredModel
    from howl_amd.training.fused import FusedTrainer
    from howl_amd.utils.synth import res8_closed_form_state, synthetic_pcm
    from oracle import frontend as ofe, models as om
    Bg, L, C = 8, 8000, 4
    pcm = synthetic_pcm(Bg, L)
    labels = torch.arange(Bg) % C
    lo, hi = parallel.shard_range(Bg, rank, world)
    with emulated_package():
        std = StandardAudioTransform().eval()
        zmuv = ZmuvTransform()
        zmuv.update(std(pcm[:2]))
        model = RegisteredModel.find_registered_class("res8")(C)
        sd0 = res8_closed_form_state(C)
        if rank != 0:      # replicas start from different weights: broadcast_parameters must make rank 0's win
            sd0 = {k: (v + 0.01 if v.is_floating_point() else v) for k, v in sd0.items()}
        model.load_state_dict(sd0, strict=False)
        model.train()
        tr = FusedTrainer(model, std, zmuv, lr=0.01, weight_decay=1e-5, late_grads=late_grads)
        tr.broadcast_parameters()
        # this rank's local gradient of step 1: the same launches outside the trainer (single-call backward), BatchNorm
        # buffers restored afterwards
        from howl_amd import ops
        bufs0 = [b.clone() for b in model.buffers()]
        feat0 = tr.features(pcm[lo:hi])
        _, dl0 = ops.xent(model._launch_forward(feat0), labels[lo:hi])
        seen = {"local": torch.cat([g.reshape(-1) for g in model._launch_backward(feat0, dl0)])}
        for b, b0 in zip(model.buffers(), bufs0):
            b.copy_(b0)
        # ... and what the optimiser is handed in step 1: the two-part backward with the all-reduce started in between
        real_adamw, calls = ops.adamw_step, []

        def spy(flat, grad, *a, **k):
            seen.setdefault("reduced", grad.clone())
            calls.append(a[-1] if a else None)
            return real_adamw(flat, grad, *a, **k)

        ops.adamw_step = spy
        try:
            for _ in range(2):
                loss = tr.step(pcm[lo:hi], labels[lo:hi])
        finally:
            ops.adamw_step = real_adamw
        weights = tr.fp.flat.clone()
        bn = torch.cat([b.reshape(-1).float() for b in model.buffers()])
    # (i) this rank's local gradient == oracle on this rank's shard
    fb = ofe.mel_fb(40)
    z = ofe.Zmuv()
    z.update(ofe.standard_audio_transform(pcm[:2], fb))
    x = z(ofe.standard_audio_transform(pcm[lo:hi], fb))
    sd = om.res8_init(C)
    names = om.res8_param_names()
    params = [sd[n].clone().requires_grad_(True) for n in names]
    sdl = dict(sd)
    sdl.update(dict(zip(names, params)))
    ref = torch.autograd.grad(torch.nn.functional.cross_entropy(om.res8_forward(sdl, x, True), labels[lo:hi]), params)
    ref = torch.cat([g.reshape(-1) for g in ref])
    shard_err = ((seen["local"] - ref).abs().max() / max(1.0, ref.abs().max().item())).item()
    locals_ = [torch.zeros_like(ref) for _ in range(world)]
    dist.all_gather(locals_, seen["local"])
    ws = [torch.zeros_like(weights) for _ in range(world)]
    dist.all_gather(ws, weights)
    bns = [torch.zeros_like(bn) for _ in range(world)]
    dist.all_gather(bns, bn)
    errs = torch.tensor([shard_err])
    dist.all_reduce(errs, op=dist.ReduceOp.MAX)
    if rank == 0:
        out["shard_err"] = errs.item()
        out["sum_err"] = (seen["reduced"] - sum(locals_)).abs().max().item()          # (ii) the reduced buffer is the sum
        out["identical"] = all(torch.equal(ws[0], w) for w in ws)                     # (iii) replicas stay identical
        out["bn_differs"] = not torch.equal(bns[0], bns[1])                           # local-batch BN statistics (no SyncBN)
        out["world"] = tr.world
        out["collectives"] = tr.collectives_last_step
    dist.destroy_process_group()


import pytest


@pytest.fixture(scope="module")
def emu_built():
    """The emulator library is built HERE, once, before any rank is spawned: two ranks finding it stale would both recompile
    every kernel source into the same object files at the same time (minutes, and a race)."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    import emu_util
    emu_util.emu_lib()


@pytest.mark.parametrize("late_grads,collectives", [("overlap", 2), ("merged", 1)])
def test_two_rank_fused_trainer_on_emulated_kernels(emu_built, late_grads, collectives):
    """(i) per-rank gradient == oracle on the rank's shard, (ii) all-reduced gradient == sum of the shard
    gradients (the 1/world mean is applied inside the AdamW kernel), (iii) bit-identical weights on all ranks after two
    steps from deliberately different initial weights (broadcast_parameters).  Both schedules of the late (conv0) gradients:
    "overlap" = two-part backward with two collectives, "merged" = one collective over the whole buffer."""
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_trainer_worker, args=(world, _free_port(), out, late_grads), nprocs=world, join=True)
        assert out["world"] == 2 and out["collectives"] == collectives
        assert out["shard_err"] < 5e-5, out["shard_err"]
        assert out["sum_err"] < 1e-6, out["sum_err"]
        assert out["identical"]
        assert out["bn_differs"]


def _entry_worker(rank, world, port, wsdir, out):
    """`python -m torch.distributed.run ... -m training.run.pretrain_gsc` as the launcher would start it: the environment
    carries the rendezvous, the entry point does the rest (kernels on the hipemu build, gloo because the tensors are on the
    host)."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), NUM_MELS="40", DEVICE="cpu", BATCH_SIZE="8", NUM_EPOCHS="1",
                      MAX_WINDOW_SIZE_SECONDS="0.5", LEARNING_RATE="0.01", SEED="3")
    torch.set_num_threads(2)
    from emu_util import emulated_package
    from howl_amd import ops
    from howl_amd.training import fused
    from howl_amd.training.run import pretrain_gsc
    shards, finals = [], []
    real_step = fused.FusedTrainer.step

    def step(self, audio, labels, *a, **k):
        shards.append(int(audio.shape[0]))
        r = real_step(self, audio, labels, *a, **k)
        finals.append(self.fp.flat.clone())
        return r

    fused.FusedTrainer.step = step
    with emulated_package():
        pretrain_gsc.main(["--model", "res8", "--workspace", wsdir, "--synthetic", "16"])
    flat = finals[-1]
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        out["steps"] = len(shards)
        out["shards"] = shards
        out["identical"] = all(torch.equal(gathered[0], g) for g in gathered)
    dist.destroy_process_group()


def test_two_rank_pretrain_gsc_entry_point(emu_built, tmp_path):
    """The entry point itself under a 2-rank launcher environment: it joins the process group, splits every global batch of 8
    into shards of 4, broadcasts rank 0's start, all-reduces inside the fused step (two-part backward), evaluates in shards,
    and only rank 0 writes the workspace; the replicas end with bit-identical weights."""
    world = 2
    wsdir = tmp_path / "ws"
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_entry_worker, args=(world, _free_port(), str(wsdir), out), nprocs=world, join=True)
        assert out["steps"] == 2 and list(out["shards"]) == [4, 4]          # 16 clips / global batch 8, half of it per rank
        assert out["identical"]
    for name in ("model.pt.bin", "model-best.pt.bin", "zmuv.pt.bin", "settings.json", "cmd-args.json"):
        assert (wsdir / name).exists(), name
    assert (wsdir / "logs" / "scalars.jsonl").exists()
