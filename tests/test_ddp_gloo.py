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
