"""Data-parallel plumbing: one process per GPU, utterances sharded by index, ONE sum all-reduce of the flat fp32
gradient buffer per step over RCCL (torch.distributed backend "nccl" on ROCm), averaged inside the AdamW kernel.

The reference has no distributed code at all; this is the only collective the hot path needs.  The buffer
is 441 KB for res8, i.e. latency-bound on xGMI, so there is no bucketing or overlap machinery -- a single call on the
compute stream right after the backward kernels.  BatchNorm uses each replica's local batch statistics (no SyncBN).
"""
import torch
import torch.distributed as dist


def world_info(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_range(global_batch: int, rank: int, world: int):
    """Contiguous utterance range [lo, hi) of `rank`; sizes differ by at most one."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_sum_(flat: torch.Tensor, group=None) -> float:
    """In-place sum over replicas; returns the scale (1/world) the optimiser applies to turn it into the mean."""
    _, world = world_info(group)
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world


def broadcast_(tensors, src=0, group=None):
    _, world = world_info(group)
    if world > 1:
        for t in tensors:
            dist.broadcast(t, src, group=group)
