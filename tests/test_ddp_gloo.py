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


def _trainer_worker(rank, world, port, out):
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
        tr = FusedTrainer(model, std, zmuv, lr=0.01, weight_decay=1e-5)
        tr.broadcast_parameters()
        # spy on the all-reduce: keep this rank's local gradient of step 1
        seen = {}
        real = parallel.allreduce_sum_

        def spy(flat, group=None):
            seen.setdefault("local", flat.clone())
            scale = real(flat, group)
            seen.setdefault("reduced", flat.clone())
            return scale

        parallel.allreduce_sum_ = spy
        try:
            for _ in range(2):
                loss = tr.step(pcm[lo:hi], labels[lo:hi])
        finally:
            parallel.allreduce_sum_ = real
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
    dist.destroy_process_group()


def test_two_rank_fused_trainer_on_emulated_kernels():
    """(i) per-rank gradient == oracle on the rank's shard, (ii) all-reduced gradient == sum of the shard
    gradients (the 1/world mean is applied inside the AdamW kernel), (iii) bit-identical weights on all ranks after two
    steps from deliberately different initial weights (broadcast_parameters)."""
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_trainer_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        assert out["world"] == 2
        assert out["shard_err"] < 5e-5, out["shard_err"]
        assert out["sum_err"] < 1e-6, out["sum_err"]
        assert out["identical"]
        assert out["bn_differs"]
