"""Data-parallel semantics on CPU (gloo, world_size 2): the all-reduced flat gradient buffers equal the mean of the
per-replica batch gradients, and an RMSprop step on them equals the single-process step on that mean."""
import importlib
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _replica_grads(rank):
    """Oracle (CPU) gradients of a small conv+IN+L1 model on this replica's shard - stands in for Trainer.step."""
    sys.path.insert(0, ROOT)
    from oracle import tfsem as T
    rng = np.random.default_rng(100)          # same weights on every replica
    w = torch.from_numpy(rng.standard_normal((3, 3, 3, 8)).astype(np.float32) * 0.2).requires_grad_(True)
    g = torch.ones(8, requires_grad=True); b = torch.zeros(8, requires_grad=True)
    data = np.random.default_rng(7).standard_normal((4, 8, 16, 3)).astype(np.float32)   # global batch of 4
    par = importlib.import_module(PKG + ".parallel")
    x = torch.from_numpy(data[par.shard_slice(4, rank, 2)])
    y = T.leaky_relu(T.instance_norm(T.conv2d(x, w, None), g, b), 0.1)
    loss = y.abs().mean()                     # a batch mean, like every loss term of train.py
    return [t.detach() for t in torch.autograd.grad(loss, (w, g, b))]


def _worker(rank, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    par = importlib.import_module(PKG + ".parallel")
    trainer = importlib.import_module(PKG + ".trainer")
    r, world, _ = par.init_from_env(backend="gloo")
    assert (r, world) == (rank, 2)
    grads = _replica_grads(rank)
    named = {"c.w": np.zeros((3, 3, 3, 8), np.float32), "n.gamma": np.ones(8, np.float32), "n.beta": np.zeros(8, np.float32),
             "n.moving_mean": np.zeros(8, np.float32)}
    fp = trainer.FlatParams(named, "cpu")
    assert fp.ntrain % 4 == 0 and "n.moving_mean" not in fp.g       # trainables first, frozen stats excluded
    for k, gr in zip(("c.w", "n.gamma", "n.beta"), grads):
        fp.g[k].copy_(gr)
    if rank == 1:
        fp.flat.add_(1.0)                                            # diverged replica: broadcast must repair it
    par.broadcast_params_([fp.flat])
    par.allreduce_sum_([fp.grad])
    torch.save({"grad": fp.grad.clone(), "flat": fp.flat.clone()}, os.path.join(out_dir, "r%d.pt" % rank))
    torch.distributed.destroy_process_group()


def test_gloo_world2_gradient_average(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), "r%d.pt" % r)) for r in (0, 1))
    assert torch.equal(r0["grad"], r1["grad"])                       # both replicas hold the same reduced buffer
    assert torch.equal(r0["flat"], r1["flat"])                       # broadcast made the weights identical
    g0, g1 = _replica_grads(0), _replica_grads(1)
    mean = [(a + b) / 2 for a, b in zip(g0, g1)]
    flat_mean = torch.cat([m.reshape(-1) for m in mean])
    got = r0["grad"][:flat_mean.numel()] / 2.0                        # gscale = 1/world is applied in the optimizer
    assert torch.allclose(got, flat_mean, rtol=1e-6, atol=1e-8)
    # and the mean of the shard gradients IS the gradient of the global-batch mean loss for per-sample-independent nets
    sys.path.insert(0, ROOT)
    from oracle import tfsem as T
    rng = np.random.default_rng(100)
    w = torch.from_numpy(rng.standard_normal((3, 3, 3, 8)).astype(np.float32) * 0.2).requires_grad_(True)
    g = torch.ones(8, requires_grad=True); b = torch.zeros(8, requires_grad=True)
    x = torch.from_numpy(np.random.default_rng(7).standard_normal((4, 8, 16, 3)).astype(np.float32))
    loss = T.leaky_relu(T.instance_norm(T.conv2d(x, w, None), g, b), 0.1).abs().mean()
    full = torch.autograd.grad(loss, (w, g, b))
    for m, f in zip(mean, full):
        assert torch.allclose(m, f, rtol=1e-4, atol=1e-7)


def test_shard_slice():
    par = importlib.import_module(PKG + ".parallel")
    assert [par.shard_slice(256, r, 8) for r in (0, 7)] == [slice(0, 32), slice(224, 256)]
    try:
        par.shard_slice(30, 0, 4)
    except ValueError:
        pass
    else:
        raise AssertionError
