"""Trainer(defer_dense=True) (round 5): the sun-pose Dense kernels' RMSprop launch opens the NEXT replay instead of closing its own
step (train.py:402-403 semantics kept: the same launches with the same arguments - only later).  Claims under test: (1) after
flush() every weight, RMSprop slot, bf16 Dense image and BatchNorm statistic equals the undeferred trainer's, bit for bit, over
several captured replays; (2) between replay() and flush() only the two Dense kernels (and their slots / images) are stale; (3)
eager step() / test_step() never leave an update pending."""
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu


def _nets():
    params = pkg("params")
    return [params.init_params(params.generator_spec(), 0), params.init_params(params.sunpose_spec(), 1),
            params.init_params(params.discriminator_spec(), 2), params.init_params(params.vgg_spec(), 3)]


def _state(tr):
    return dict(gflat=tr.gs.flat.clone(), gms=tr.gs.ms.clone(), dflat=tr.ds.flat.clone(), dms=tr.ds.ms.clone(),
                fc1pk=tr.fc1.pk_hi.clone(), fc1nat=tr.fc1.nat_hi.clone(), fc2pk=tr.fc2.pk_hi.clone(), fc2nat=tr.fc2.nat_hi.clone())


def test_deferred_dense_update_equals_the_undeferred_step(dev):
    synth, trainer, K = pkg("synth"), pkg("trainer"), pkg("kernels")
    nets = _nets()
    B = 4
    batches = [synth.make_batch(B, seed=300 + i) for i in range(3)]
    dv = lambda b: [torch.from_numpy(b[k]).to(dev) for k in ("ldr", "hdr_t", "sunpose_gt")]
    ta = trainer.Trainer(*nets, device=dev, precise=False, compute=K.BF16, lr=2e-6)
    tb = trainer.Trainer(*nets, device=dev, precise=False, compute=K.BF16, lr=2e-6, defer_dense=True)
    assert tb._defer and not ta._defer
    bufs_a, bufs_b = dv(batches[0]), dv(batches[0])
    ta.capture(*bufs_a); tb.capture(*bufs_b)
    assert not tb._fc_pending
    for k in _state(ta):
        assert torch.equal(_state(ta)[k], _state(tb)[k]), "state after capture: %s" % k
    o, n, _ = tb.gs.offsets["sun.fc1.kernel"]
    for it, b in enumerate(batches):
        for dst_a, dst_b, src in zip(bufs_a, bufs_b, dv(b)):
            dst_a.copy_(src); dst_b.copy_(src)
        prev_a = ta.gs.flat[o:o + n].clone()      # the undeferred trainer's first Dense kernel after the PREVIOUS step
        ta.replay(); tb.replay()
        torch.cuda.synchronize()
        assert tb._fc_pending
        sa, sb = _state(ta), _state(tb)
        # (2) conv-side parameters of both optimizers are current, the first Dense kernel is what it was before THIS step's update
        fc0 = tb.fc_grad_range()[0]
        assert torch.equal(sa["gflat"][:fc0], sb["gflat"][:fc0]) and torch.equal(sa["dflat"], sb["dflat"]), it
        # (RMSprop's first steps are sign-like: lr is kept small so that the Dense layers stay alive and every step moves them)
        assert not torch.equal(sa["gflat"][o:o + n], sb["gflat"][o:o + n]) and torch.equal(sb["gflat"][o:o + n], prev_a), it
        if it == 1:      # a flush in the middle of a run: nothing pending afterwards, the next replay starts without the late segment
            tb.flush(); torch.cuda.synchronize()
            assert not tb._fc_pending
            for k in sa:
                assert torch.equal(sa[k], _state(tb)[k]), "after flush, step %d: %s" % (it, k)
    tb.flush(); torch.cuda.synchronize()
    sa, sb = _state(ta), _state(tb)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), "after the last flush: %s (%d elements differ)" % (k, int((sa[k] != sb[k]).sum()))
    assert torch.equal(ta.losses, tb.losses) or torch.allclose(ta.losses, tb.losses, rtol=1e-5)
    # (3) eager passes flush themselves
    ta.step(*dv(batches[0])); tb.step(*dv(batches[0])); torch.cuda.synchronize()
    assert not tb._fc_pending
    for k in _state(ta):
        assert torch.equal(_state(ta)[k], _state(tb)[k]), "after an eager step: %s" % k
    tb.replay(); assert tb._fc_pending
    tb.test_step(*dv(batches[1])); assert not tb._fc_pending
    # a gradient-only replay consumes a pending update and leaves none
    tb.replay(); tb.replay(update=False); torch.cuda.synchronize()
    assert not tb._fc_pending
