"""Data-parallel plumbing: one process per GPU, utterances sharded by index, sum all-reduces of the flat fp32 gradient
buffer over RCCL (torch.distributed backend "nccl" on ROCm), averaged inside the AdamW kernel.

The reference has no distributed code at all; this is the only collective the hot path needs.  The buffer is
441 KB for res8, i.e. latency-bound on xGMI, so there is no bucketing.  torch.distributed runs a collective on the process
group's own stream, ordered after the work already queued on the current (compute) stream; ``allreduce_start_`` returns the
handle so that the caller can keep launching kernels under it (the fused res8 step overlaps conv0's weight gradient with the
all-reduce of everything else) and ``wait`` orders the compute stream behind it again.  BatchNorm uses each replica's local
batch statistics (no SyncBN).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(device: torch.device = None):
    """Join the job described by the launcher's environment (``torchrun`` / ``python -m torch.distributed.run``: RANK,
    LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT).  One process per GPU: a HIP ``device`` selects RCCL ("nccl") and is
    bound to the process group, anything else (CPU tests) gloo.  Returns (rank, world, device); without WORLD_SIZE > 1 in the
    environment nothing is initialised and (0, 1, device) comes back."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, device
    rank, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size(), device
    if device is not None and device.type == "cuda":
        n = torch.cuda.device_count()
        if local_rank >= n:
            raise RuntimeError(f"rank {rank}: LOCAL_RANK={local_rank} but only {n} HIP device(s) are visible (one process per GPU)")
        device = torch.device("cuda", local_rank)
        torch.cuda.set_device(device)
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group("gloo")
    if dist.get_world_size() != world:
        raise RuntimeError(f"process group has {dist.get_world_size()} ranks, the launcher announced {world}")
    return dist.get_rank(), dist.get_world_size(), device


def world_info(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def is_main(group=None) -> bool:
    return world_info(group)[0] == 0


def barrier(group=None):
    if world_info(group)[1] > 1:
        dist.barrier(group=group)


def shard_range(global_batch: int, rank: int, world: int):
    """Contiguous utterance range [lo, hi) of `rank`; sizes differ by at most one."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(items, group=None):
    """This rank's contiguous share of one global batch (a list of clip ids / examples)."""
    rank, world = world_info(group)
    lo, hi = shard_range(len(items), rank, world)
    return items[lo:hi]


def allreduce_sum_(flat: torch.Tensor, group=None) -> float:
    """In-place sum over replicas; returns the scale (1/world) the optimiser applies to turn it into the mean."""
    _, world = world_info(group)
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world


class _Done:
    def wait(self):
        return True


def allreduce_start_(flat: torch.Tensor, group=None):
    """Start the in-place sum of ``flat`` (a contiguous view) and return a handle with ``wait()``; kernels launched on the
    compute stream before ``wait()`` run under the collective."""
    _, world = world_info(group)
    if world > 1 and flat.numel():
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return _Done()


def allreduce_scalars_(values: torch.Tensor, group=None) -> torch.Tensor:
    """Sum of a small tensor of counters over the replicas (sharded evaluation)."""
    if world_info(group)[1] > 1:
        dist.all_reduce(values, op=dist.ReduceOp.SUM, group=group)
    return values


def broadcast_(tensors, src=0, group=None):
    _, world = world_info(group)
    if world > 1:
        for t in tensors:
            dist.broadcast(t, src, group=group)
