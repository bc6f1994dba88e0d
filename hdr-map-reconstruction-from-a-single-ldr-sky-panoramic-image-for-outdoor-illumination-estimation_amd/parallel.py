"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI on ROCm;
"gloo" on CPU for tests).  The reference is single-GPU (train.py:19-24); the semantics chosen for N replicas
(SURVEY.md section 8e): every replica runs the reference's batch-32 step on its own shard - local BatchNorm
statistics, local reduce_max - and the gradients are averaged, which is exact because every loss term is a batch
mean (train.py:305-331,364-369).

The exchange is over the two FLAT gradient buffers (one per optimizer, trainer.FlatParams) - two large collectives
per step instead of ~190 small ones; the 1/world factor is folded into the optimizer kernel (`gscale`).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, device=None):
    """RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment (torchrun contract).  Returns (rank, world, local)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_slice(global_batch, rank, world):
    """Replica r owns images [r*B/world, (r+1)*B/world) of the global batch."""
    per = global_batch // world
    if per * world != global_batch:
        raise ValueError("global batch %d is not divisible by world size %d" % (global_batch, world))
    return slice(rank * per, (rank + 1) * per)


def allreduce_sum_(flat_buffers):
    """In-place SUM over replicas of each flat gradient buffer (the mean's 1/world goes into the optimizer)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in flat_buffers:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


def broadcast_params_(flat_buffers, src=0):
    """Replicas start from identical weights (rank `src`'s)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in flat_buffers:
        dist.broadcast(t, src=src)
