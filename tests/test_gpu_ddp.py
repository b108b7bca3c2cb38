"""-m gpu coverage of the data-parallel path: the product's FusedTrainer under torch.distributed on device
tensors.  Two RCCL ranks when the box has two GPUs; two gloo ranks sharing ONE GPU otherwise (RCCL refuses two ranks per
device) -- the control flow (parameter broadcast, in-step all-reduce of the flat gradient, rank-0 reporting, the roofline pass
on every rank) is the same either way."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, backend, out):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NUM_MELS="40", HSA_ENABLE_IPC_MODE_LEGACY="0")
    ndev = torch.cuda.device_count()
    dev = torch.device(f"cuda:{rank % ndev}")
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from howl_amd import parallel
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    from howl_amd.model import RegisteredModel
    from howl_amd.training.fused import FusedTrainer
    from howl_amd.utils.synth import res8_closed_form_state, synthetic_pcm
    from oracle import frontend as ofe, models as om
    Bg, L, C = 64, 16000, 12
    pcm = synthetic_pcm(Bg, L)
    labels = torch.arange(Bg) % C
    lo, hi = parallel.shard_range(Bg, rank, world)
    std = StandardAudioTransform().to(dev).eval()
    zmuv = ZmuvTransform().to(dev)
    zmuv.update(std(pcm[:4].to(dev)))
    model = RegisteredModel.find_registered_class("res8")(C)
    sd0 = res8_closed_form_state(C)
    if rank != 0:          # broadcast_parameters must make rank 0's weights win
        sd0 = {k: (v + 0.01 if v.is_floating_point() else v) for k, v in sd0.items()}
    model.load_state_dict(sd0, strict=False)
    model = model.to(dev).train()
    tr = FusedTrainer(model, std, zmuv, lr=0.01, weight_decay=1e-5)
    tr.broadcast_parameters()
    # this rank's local gradient of step 1 (single-call backward outside the trainer; BatchNorm buffers restored) ...
    from howl_amd import ops
    bufs0 = [b.clone() for b in model.buffers()]
    feat0 = tr.features(pcm[lo:hi].to(dev))
    _, dl0 = ops.xent(model._launch_forward(feat0), labels[lo:hi].to(dev))
    seen = {"local": torch.cat([g.reshape(-1) for g in model._launch_backward(feat0, dl0)])}
    for b, b0 in zip(model.buffers(), bufs0):
        b.copy_(b0)
    # ... and what the optimiser is handed in step 1 (two-part backward, all-reduce started in between)
    real_adamw = ops.adamw_step

    def spy(flat, grad, *a, **k):
        seen.setdefault("reduced", grad.clone())
        return real_adamw(flat, grad, *a, **k)

    ops.adamw_step = spy
    for _ in range(3):
        tr.step(pcm[lo:hi].to(dev), labels[lo:hi].to(dev))
    ops.adamw_step = real_adamw
    torch.cuda.synchronize()
    # (i) local gradient == oracle on this rank's shard
    fb = ofe.mel_fb(40)
    z = ofe.Zmuv()
    z.update(ofe.standard_audio_transform(pcm[:4], fb))
    x = z(ofe.standard_audio_transform(pcm[lo:hi], fb))
    sd = om.res8_init(C)
    names = om.res8_param_names()
    params = [sd[n].clone().requires_grad_(True) for n in names]
    sdl = dict(sd)
    sdl.update(dict(zip(names, params)))
    ref = torch.autograd.grad(torch.nn.functional.cross_entropy(om.res8_forward(sdl, x, True), labels[lo:hi]), params)
    ref = torch.cat([g.reshape(-1) for g in ref])
    local = seen["local"].cpu()
    shard_err = ((local - ref).abs().max() / max(1.0, ref.abs().max().item())).item()   # tolerance of test_gpu_res8
    gather = lambda t: [torch.zeros_like(t) for _ in range(world)]
    locals_, ws = gather(seen["local"]), gather(tr.fp.flat)
    dist.all_gather(locals_, seen["local"])
    dist.all_gather(ws, tr.fp.flat)
    errs = torch.tensor([shard_err], device=dev)
    dist.all_reduce(errs, op=dist.ReduceOp.MAX)
    if rank == 0:
        out["shard_err"] = errs.item()
        tot = sum(l.double() for l in locals_)
        out["sum_err"] = ((seen["reduced"].double() - tot).abs().max() / tot.abs().max()).item()
        out["identical"] = all(torch.equal(ws[0], w) for w in ws)
        out["world"] = tr.world
        out["backend"] = dist.get_backend()
    dist.destroy_process_group()


def _rccl_single_rank(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    sys.path.insert(0, str(ROOT))
    from howl_amd import parallel
    flat = torch.arange(110307, dtype=torch.float32, device=dev)            # res8's flat gradient buffer
    want = flat.clone()
    scale = parallel.allreduce_sum_(flat)                                    # dist.all_reduce on the process group's stream
    pending = parallel.allreduce_start_(flat[64:])                           # the async form of the overlap schedule
    pending.wait()
    parallel.broadcast_([flat, torch.zeros(3, dtype=torch.int64, device=dev)])
    parallel.barrier()
    counts = parallel.allreduce_scalars_(torch.tensor([3.0, 5.0], device=dev))
    torch.cuda.synchronize()
    out["ok"] = bool(torch.equal(flat, want)) and scale == 1.0 and counts.tolist() == [3.0, 5.0]
    out["backend"] = dist.get_backend()
    dist.destroy_process_group()


def test_rccl_communicator_on_this_box():
    """RCCL itself on the GPU box: a one-rank "nccl" process group (communicator creation, all-reduce -- blocking and async --,
    broadcast, barrier) through howl_amd.parallel's helpers.  It cannot show scaling; it does show that the RCCL library loads
    and runs collectives on this software stack with the environment the benches set (HSA_ENABLE_IPC_MODE_LEGACY=0)."""
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_rccl_single_rank, args=(1, _free_port(), out), nprocs=1, join=True)
        out = dict(out)
    assert out["backend"] == "nccl" and out["ok"], out


def _run(world, backend):
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), backend, out), nprocs=world, join=True)
        return dict(out)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs for RCCL (one rank per device)")
def test_fused_trainer_two_rccl_ranks():
    out = _run(2, "nccl")
    assert out["world"] == 2 and out["backend"] == "nccl"
    assert out["shard_err"] < 5e-5 and out["sum_err"] < 1e-6 and out["identical"], out


def test_fused_trainer_two_gloo_ranks_on_one_gpu():
    """The same trainer code with device tensors and two processes on this box's GPU(s): per-shard gradients vs the oracle,
    the reduced buffer is the sum of the shard gradients, replicas bit-identical after three steps."""
    out = _run(2, "gloo")
    assert out["world"] == 2
    assert out["shard_err"] < 5e-5 and out["sum_err"] < 1e-6 and out["identical"], out


def test_bench_self_spawns_ranks():
    """`python bench.py --gpus 2` with no distributed environment starts two ranks itself and reports n_gpus = 2 (gloo on one
    GPU here; the driver's multi-GPU run uses RCCL, one rank per device)."""
    env = dict(os.environ, HOWL_BENCH_BACKEND="gloo" if torch.cuda.device_count() < 2 else "nccl")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                          "--batch-per-gpu", "64"], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["config"]["global_batch"] == 128
    assert line["rccl"]["world_size"] == 2 and line["rccl"]["allreduce_bytes"] == 110307 * 4
    assert line["roofline"]["launches"] == 6 * 4 and line["cpu_baseline"] is None
    assert line["value"] > 0 and line["scaling"] == "weak"
