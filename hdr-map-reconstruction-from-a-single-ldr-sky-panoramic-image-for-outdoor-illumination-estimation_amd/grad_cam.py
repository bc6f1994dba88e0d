"""Mirror of `grad_cam.layer` (grad_cam.py:29-44) without a tape.

The reference asks TensorFlow for d y_c / d A_k three times (`tf.gradients`, graph mode).  Here the sun-pose net
records what its backward sweep needs while it runs, `pick()` defines y_c, and the first `layer()` call runs ONE
shared sweep for all three activation maps (engine.gradcam_sweep); later calls return the cached maps.

    cmf, (A1, A2, A3) = sun.sunposeEstimation(x, training=False)
    y_c  = grad_cam.pick(cmf)            # inference.py:98   reduce_max(cmf, axis=1)
    y_c  = grad_cam.pick(cmf, gt)        # train.py:265-267  cmf[b, argmax gt[b]]
    cam1 = grad_cam.layer(y_c, A1)       # [B,H,W,1]
"""
import torch


class PickedProbability:
    """y_c [B] plus the graph context `layer` differentiates through."""

    def __init__(self, values, ctx, pick_src):
        self.values, self._ctx, self._pick_src, self._cams = values, ctx, pick_src, None

    def __array__(self):
        return self.values.cpu().numpy()


def pick(cmf, sunpose_gt=None):
    ctx = getattr(cmf, "_hdrsky_ctx", None)
    if ctx is None:
        raise ValueError("cmf does not come from sunpose_net.model.sunposeEstimation (no recorded graph)")
    src = cmf if sunpose_gt is None else sunpose_gt
    idx = src.argmax(dim=1, keepdim=True)
    return PickedProbability(cmf.gather(1, idx).squeeze(1), ctx, src)


def layer(y_c, A_k):
    if not isinstance(y_c, PickedProbability):
        raise TypeError("y_c must come from grad_cam.pick(): a bare tensor carries no graph to differentiate")
    model, t = y_c._ctx
    if y_c._cams is None:
        from . import engine
        y_c._cams = engine.gradcam_sweep(model.nets, t, y_c._pick_src, model.compute)
    for k in (1, 2, 3):
        if A_k is t["A%d" % k]:
            return y_c._cams[k - 1]
    raise ValueError("A_k is not one of the activation maps returned with this cmf")
