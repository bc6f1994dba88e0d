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

from . import hooks as HOOKS


def init_from_env(backend=None, device=None):
    """RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment (torchrun contract).  Returns (rank, world, local)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # HDRSKY_DIST_BACKEND=gloo: rehearsal of the multi-process path with several ranks on ONE card (RCCL refuses that)
        backend = backend or HOOKS.H.dist_backend or ("nccl" if torch.cuda.is_available() else "gloo")
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


class BatchSync:
    """The in-step collectives of the optional "one global batch" semantics: with it N replicas of
    batch B compute the reference's step at batch N*B exactly - Keras BatchNormalization statistics (discriminator.py:16,24,
    sunrad_net.py:17,25) and tf.reduce_max(sunpose_pred) (generator.py:160) run over the batch of EVERY replica, forward
    and backward - instead of the default N independent batch-B steps whose gradients are averaged.  The trainer calls these
    at the four coupling points; every call is a small collective on the step's own stream (a few KB: per-tile moment
    partials, partial gradient sums, one word), so such a step is issued eagerly, not replayed from hipGraphs."""

    def __init__(self):
        self.world = dist.get_world_size() if dist.is_initialized() else 1

    def gather_rows(self, t):
        """[n, ...] -> [world * n, ...]: the rows of every replica, in replica order (identical on every replica)."""
        t = t.contiguous()
        if self.world == 1:
            return t
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t)
        return out

    def gather_stats(self, st):
        """Per-tile moment partials [B, nparts, 2, C] of a conv output -> those of the global batch [world * B, ...]."""
        from .kernels import Stats
        return Stats(self.gather_rows(st.part), st.nparts, st.count)

    def max_word(self, w):
        """int32[1] holding the bit pattern of a non-negative float (the soft-max head's max accumulator): max over replicas."""
        if self.world > 1:
            dist.all_reduce(w, op=dist.ReduceOp.MAX)
        return w


MODES = ("allreduce", "allreduce_bf16", "gather_dense")
DEFAULT_MODE = "gather_dense"      # 53 MB per step instead of 233 MB, and the Dense gradients are never even written


class GradientExchange:
    """The per-step gradient exchange of a data-parallel `trainer.Trainer`, as hooks for `Trainer.replay`.

    What has to travel per step and replica (32x128, fp32): the two sun-pose Dense kernels 201 MB, everything else of the
    generator / sun-pose optimizer 21 MB, the discriminator 11 MB.  xGMI is point-to-point (7 links x ~153 GB/s per GPU),
    so a ring all-reduce of S bytes costs about 2 (N-1)/N S / (one link's ~300 GB/s both ways); modes (`mode=` or the
    environment variable HDRSKY_DP_MODE):

      (default: gather_dense)
      allreduce       three flat all-reduces (SUM; the 1/world goes into the optimizer kernel).  The Dense slice (201 MB)
                      is complete after segment `Trainer.FC_GRADS_READY`: its collective is enqueued by the host late (as
                      a pre-hook of the Dense-layer optimizer segment, so the enqueue cost does not sit between the
                      launches of the critical chain) but on a communication stream that waits only for that segment's
                      event - on the GPU it starts as soon as those gradients exist and runs beside the remaining ~60 % of
                      the backward pass.  The other 32 MB are reduced when `Trainer.GRADS_READY` is reached.
                      233 MB per step; ring at N=8: ~1.4 ms, of which 86 % is overlapped.
      allreduce_bf16  the same schedule with a bf16 payload (the gradient slices are rounded to bf16, summed by RCCL in
                      bf16 and widened again): 116 MB per step.  Changes the arithmetic (2^-9 relative per term and per
                      addition): offered for bandwidth-starved topologies, not the default.
      gather_dense    the Dense kernels' gradients never travel.  A Dense weight gradient is X^T dY; the replicas
                      all-gather X and dY instead - flat [B,8192], df1 [B,4096], f1 [B,4096], dz [B,4096], 2.6 MB per
                      replica, 21 MB gathered at N=8 - and every replica recomputes the FULL-batch weight gradients of
                      fc1 / fc2 (its local Dense weight-gradient launches are skipped: `Trainer.dense_wgrad_external`).
                      The sum over replicas is the same number the all-reduce produces (fp32 summation order differs).
                      With `Trainer.fused_dense` (HDRSKY_BF16 mode) that gradient is not even written: the Dense
                      optimizer launch contracts the gathered rows tile by tile (hdrsky_rmsprop_fc_fused, M = N * B).
                      32 MB all-reduced + 21 MB gathered per step at N=8 instead of 233 MB.

        ex = GradientExchange(tr, mode="gather_dense")
        tr.replay(hooks=ex.hooks, pre_hooks=ex.pre_hooks)      # graph path
        tr.step(..., update=False); ex.reduce_all(); tr.apply_gradients()   # eager path
    """

    def __init__(self, trainer, device=None, mode=None, sync_batch_stats=False):
        """sync_batch_stats=True: BatchNorm batch statistics and the batch-global maximum run over every replica's batch
        (BatchSync) - N replicas of batch B then take the reference's batch N*B step, to fp32 round-off.  Such a step couples
        the replicas inside the forward and backward passes, so it is issued eagerly (step / reduce_all /
        apply_gradients), not captured."""
        mode = mode or HOOKS.H.dp_mode or DEFAULT_MODE
        if mode not in MODES:
            raise ValueError("unknown data-parallel mode %r (one of %s)" % (mode, ", ".join(MODES)))
        self.tr, self.mode = trainer, mode
        self.fc0, self.fc1 = trainer.fc_grad_range()
        self.active = dist.is_initialized()     # a process group exists (world 1 only in rehearsals of this path)
        self.world = dist.get_world_size() if self.active else 1
        self.comm = torch.cuda.Stream(device=device) if self.active else None
        if getattr(trainer, "_graphs", None) is not None:
            raise RuntimeError("GradientExchange: construct before Trainer.capture (it changes which launches the plan holds)")
        self.has_dense = self.fc1 > self.fc0    # False for a sunpose="external" trainer: nothing but the conv slices travels
        if not self.has_dense:
            pass
        elif mode == "gather_dense" and self.active:
            trainer.dense_wgrad_external = True   # the local Dense weight-gradient launches are left out (see _dense_gather)
            trainer.on_bind = self._on_bind
        elif self.active:
            trainer.fused_dense = False           # the Dense gradients travel: they have to be written out
        self._gbuf = {}
        self.sync_batch_stats = bool(sync_batch_stats)
        if self.sync_batch_stats:
            trainer.sync = BatchSync()

    def describe(self):
        return {"allreduce": "RCCL all-reduce of fp32 gradients, Dense slice overlapped with backward",
                "allreduce_bf16": "RCCL all-reduce of bf16-rounded gradients, Dense slice overlapped with backward",
                "gather_dense": "Dense gradients recomputed from all-gathered activations; RCCL all-reduce of the rest"}[self.mode]

    # ---- payload helpers ------------------------------------------------------------------------------------------
    def _allreduce(self, t):
        if self.mode == "allreduce_bf16":
            key = (t.data_ptr(), t.numel())
            buf = self._gbuf.get(key)
            if buf is None:
                buf = self._gbuf[key] = torch.empty(t.numel(), dtype=torch.bfloat16, device=t.device)
            buf.copy_(t)
            dist.all_reduce(buf)
            t.copy_(buf)
            return None
        dist.all_reduce(t)
        return None

    def _dense_gather(self):
        """all-gather (flat | df1 | f1 | dz) of every replica and recompute both Dense weight (and bias) gradients on the
        global batch, in place of the local ones."""
        from . import kernels as K
        tr = self.tr
        T, g = tr._T, tr.gs.g
        parts = [T["t"]["flat"], T["df1"], T["t"]["f1"], T["dz"]]
        local, allp, views = self._dense_buffers(parts[0].shape[0])
        K.concat_rows4(parts, local)           # one libhdrsky launch (no torch-owned kernel in the step)
        dist.all_gather_into_tensor(allp, local)
        if tr.fused_dense:
            return     # the Dense optimizer launch contracts the gathered rows itself (Trainer.dense_operands = views)
        flat, df1, f1, dz = views
        if not tr.dense_mfma:
            K.fc_wgrad(f1.contiguous(), dz.contiguous(), g["sun.fc2.kernel"], g["sun.fc2.bias"])
            K.fc_wgrad(flat.contiguous(), df1.contiguous(), g["sun.fc1.kernel"], g["sun.fc1.bias"])
        else:
            K.fc_wgrad_bf16(f1, dz, g["sun.fc2.kernel"], g["sun.fc2.bias"])
            K.fc_wgrad_bf16(flat, df1, g["sun.fc1.kernel"], g["sun.fc1.bias"])

    def _dense_buffers(self, B):
        """Static buffers of the operand exchange for a per-replica batch of B: (local [B, W], gathered [world*B, W],
        the four column blocks flat | df1 | f1 | dz of the gathered one - row-strided views)."""
        st = self._gbuf.get(("dense", B))
        if st is None:
            tr = self.tr
            widths = [tr.fc1.K, tr.fc1.N, tr.fc2.K, tr.fc2.N]
            dev = tr.gs.flat.device
            # zeros, not empty: Trainer.capture's warm-up steps run without the exchange hooks, i.e. the fused Dense update
            # contracts these rows before any all-gather has filled them (its effect is undone afterwards - it must be finite)
            local = torch.zeros((B, sum(widths)), dtype=torch.float32, device=dev)
            allp = torch.zeros((self.world * B, sum(widths)), dtype=torch.float32, device=dev)
            o = [0]
            for wd in widths:
                o.append(o[-1] + wd)
            st = self._gbuf[("dense", B)] = (local, allp, tuple(allp[:, o[i]:o[i + 1]] for i in range(4)))
        return st

    def _on_bind(self, B):
        # the Dense optimizer launch of a fused_dense trainer reads the gathered rows: they have to be known when the
        # plan is built (and captured), not only when the first gather has run
        self.tr.dense_operands = self._dense_buffers(B)[2] if self.tr.fused_dense else None

    # ---- the hook points ---------------------------------------------------------------------------------------------
    # Every collective is enqueued on ONE communication stream, in plan order (identical on every rank): the
    # discriminator slice behind `disc_step` (it is complete ~1 ms before the end of the backward pass), the Dense slice
    # / operands behind FC_GRADS_READY, the remaining conv slice behind GRADS_READY.  The compute streams only wait where
    # they consume the result (apply_fc: the Dense slice; apply: everything).
    def _on_comm(self, fn, after=None):
        """Runs fn() (collectives) on the communication stream, ordered behind `after` (an event) or - default - behind
        everything the current stream has been told to wait for so far."""
        if after is None:
            after = torch.cuda.Event()
            after.record(torch.cuda.current_stream())
        self.comm.wait_event(after)
        with torch.cuda.stream(self.comm):
            fn()

    def fc_grads_reduce(self):
        """Dense-layer slice: runs on the communication stream from the moment the slice (or, gather_dense, its operands)
        exists; the CURRENT stream (the Dense optimizer segment's) waits for it."""
        tr = self.tr
        if self.mode == "gather_dense":
            self._on_comm(self._dense_gather, after=tr.event(tr.FC_GRADS_READY))
        else:
            self._on_comm(lambda: self._allreduce(tr.gs.grad[self.fc0:self.fc1]), after=tr.event(tr.FC_GRADS_READY))
        torch.cuda.current_stream().wait_stream(self.comm)

    def disc_grads_reduce(self):
        """Hook behind `disc_step` (its weight gradients are flushed inside the segment): the discriminator's 11 MB travel
        while the generator's backward pass is still running."""
        self._on_comm(lambda: self._allreduce(self.tr.ds.grad))

    def grads_ready(self):
        """Hook of the (empty) GRADS_READY segment, whose stream has waited for every gradient producer: the conv slice of
        the generator / sun-pose optimizer; the optimizer segment behind it waits for the communication stream."""
        self._on_comm(lambda: self._allreduce(self.tr.gs.grad[:self.fc0]))
        torch.cuda.current_stream().wait_stream(self.comm)

    def reduce_all(self):
        """Eager path: every gradient buffer, on the current stream."""
        if not self.active:
            return
        if self.mode == "gather_dense":
            if self.has_dense:
                self._dense_gather()
            self._allreduce(self.tr.gs.grad[:self.fc0]); self._allreduce(self.tr.ds.grad)
        elif self.mode == "allreduce_bf16":
            self._allreduce(self.tr.gs.grad); self._allreduce(self.tr.ds.grad)
        else:
            allreduce_sum_([self.tr.gs.grad, self.tr.ds.grad])

    @property
    def hooks(self):
        return {self.tr.DISC_GRADS_READY: self.disc_grads_reduce, self.tr.GRADS_READY: self.grads_ready} if self.active else None

    @property
    def pre_hooks(self):
        return {self.tr.APPLY[0]: self.fc_grads_reduce} if self.active and self.has_dense else None


def sync_moving_stats_(trainer, src=0):
    """BatchNorm moving statistics (discriminator, sun-radiance head) are replica-local by construction (SURVEY.md section
    8e: local BN statistics).  Before a checkpoint or a validation pass every replica takes rank `src`'s copy, so that
    what is saved / evaluated does not depend on which replica happens to write or on its data shard."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for fp in (trainer.gs, trainer.ds):
        if fp.flat.numel() > fp.ntrain:
            dist.broadcast(fp.flat[fp.ntrain:], src=src)
