"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI on ROCm;
"gloo" on CPU for tests).  The reference is single-GPU (train.py:19-24); the semantics chosen for N replicas
(SURVEY.md section 8e): every replica runs the reference's batch-32 step on its own shard - local BatchNorm
statistics, local reduce_max - and the gradients are averaged, which is exact because every loss term is a batch
mean (train.py:305-331,364-369).

The exchange is over the two FLAT gradient buffers (one per optimizer, trainer.FlatParams) - three large collectives
per step instead of ~190 small ones; the 1/world factor is folded into the optimizer kernel (`gscale`).
`GradientExchange` attaches them to the Trainer's segment plan so that the largest one overlaps the backward pass.
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
        # HDRSKY_DIST_BACKEND=gloo: rehearsal of the multi-process path with several ranks on ONE card (RCCL refuses that)
        backend = backend or os.environ.get("HDRSKY_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
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


class GradientExchange:
    """The per-step gradient exchange of a data-parallel `trainer.Trainer`, as hooks for `Trainer.replay`.

    The sun-pose Dense weight gradients (201 of the 233 MB) are complete after the segment `Trainer.FC_GRADS_READY`;
    their all-reduce is enqueued by the host late (as a pre-hook of the Dense-layer optimizer segment, so the
    collective's enqueue cost does not sit between the launches of the critical chain) but on a communication stream
    that waits only for that segment's event: on the GPU it starts as soon as those gradients exist and runs beside
    the rest of the backward pass.  Everything else (32 MB) is reduced when `Trainer.GRADS_READY` is reached.

        ex = GradientExchange(tr)
        tr.replay(hooks=ex.hooks, pre_hooks=ex.pre_hooks)      # graph path
        tr.step(..., update=False); ex.reduce_all(); tr.apply_gradients()   # eager path
    """

    def __init__(self, trainer, device=None):
        self.tr = trainer
        self.fc0, self.fc1 = trainer.fc_grad_range()
        self.active = dist.is_initialized()     # a process group exists (world 1 only in rehearsals of this path)
        self.comm = torch.cuda.Stream(device=device) if self.active else None

    def fc_grads_reduce(self):
        tr = self.tr
        self.comm.wait_event(tr.event(tr.FC_GRADS_READY))
        with torch.cuda.stream(self.comm):
            work = dist.all_reduce(tr.gs.grad[self.fc0:self.fc1], async_op=True)
        work.wait()          # the CURRENT stream (the optimizer segment's) waits for the collective

    def grads_ready(self):
        dist.all_reduce(self.tr.gs.grad[:self.fc0])
        dist.all_reduce(self.tr.ds.grad)

    def reduce_all(self):
        """Eager path: every gradient buffer, on the current stream."""
        allreduce_sum_([self.tr.gs.grad, self.tr.ds.grad])

    @property
    def hooks(self):
        return {self.tr.GRADS_READY: self.grads_ready} if self.active else None

    @property
    def pre_hooks(self):
        return {self.tr.APPLY[0]: self.fc_grads_reduce} if self.active else None
