"""`python -m <pkg>.train_sun` - the reference's sun-pose pre-training CLI (train_sun.py:474-491) on the MI355X step.

Flags keep the reference's names and defaults: --dir --train --inference_img_dir --lr(1e-4) --batchsize(32)
--epochs(1000) --imheight(32) --imwidth(128) --dorf.  The reference script does not run as committed (it reads
`utils.utils.str2bool` and `args.dorfpath`, neither of which exists: train_sun.py:155,167); what is mirrored is its
training step (train_sun.py:220-264: KL + DoG loss on the sun-position map, Adam) and its checkpoint surface
(`checkpoints/SUN`, Checkpoint(epoch, lin, optimizer), max_to_keep=5, every 10th epoch: train_sun.py:186-198,345-356).
Without the Laval dataset the loop runs on seeded synthetic batches.  Launch with torchrun for data parallelism.
"""
import argparse
import os
import time

import torch

from . import checkpoint as ckpt
from . import kernels as K
from . import parallel as par
from . import params as P
from . import synth
from .trainer import SunPoseTrainer


def str2bool(v):
    return str(v).lower() in ("1", "true", "t", "yes", "y")


def main(argv=None):
    cwd = os.getcwd()
    ap = argparse.ArgumentParser(description="pretraining sun luminance estimator")
    ap.add_argument("--dir", type=str, default=None, help="dataset directory (TFRecords; not supported here -> synthetic)")
    ap.add_argument("--train", type=str, default="true")
    ap.add_argument("--inference_img_dir", type=str, default=None)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--batchsize", type=int, default=32)
    ap.add_argument("--epochs", type=int, default=1000)
    ap.add_argument("--imheight", type=int, default=32)
    ap.add_argument("--imwidth", type=int, default=128)
    ap.add_argument("--dorf", type=str, default=None,
                    help="dorfCurves.txt (utils.getDoRF): camera response curves of the LDR synthesis; default: a 1/2.2 gamma")
    ap.add_argument("--sun", type=str, default=os.path.join(cwd, "checkpoints/SUN"))
    ap.add_argument("--steps-per-epoch", type=int, default=8, help="synthetic mode: steps per epoch")
    ap.add_argument("--distortion-aware", action="store_true",
                    help="distortion_aware_ops.conv2d in every sunposeLayer (the lines sunpose_net.py:11,16 keep commented out)")
    ap.add_argument("--host-synth", action="store_true",
                    help="build the synthetic batches with numpy on the host (40 ms per batch of 32) instead of on the GPU")
    args = ap.parse_args(argv)

    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    rank, world, _ = par.init_from_env(device=dev)
    h, w = args.imheight, args.imwidth
    crf_train = None
    if args.dorf:                    # utils.py:105-116 (the script's `args.dorfpath`, train_sun.py:167, is this flag)
        if args.host_synth:
            raise SystemExit("--dorf needs the device-side LDR synthesis (drop --host-synth)")
        crf_train = synth.load_dorf(args.dorf)[0]
    sun = P.init_params(P.sunpose_spec(h, w), 1)
    mgr = ckpt.CheckpointManager(args.sun)
    tensors, epoch0 = mgr.restore()
    if tensors:
        print("Latest checkpoint has restored!! (%d variables)" % ckpt.load_into(sun, tensors, "lin"))
    tr = SunPoseTrainer(sun, device=dev, lr=args.lr, im_height=h, im_width=w, compute=K.BF16, world_size=world,
                        distortion_aware=args.distortion_aware)
    if tensors and "optimizer/m" in tensors:
        tr.adam_m.copy_(torch.from_numpy(tensors["optimizer/m"])); tr.adam_v.copy_(torch.from_numpy(tensors["optimizer/v"]))
        tr.steps_done = int(tensors["optimizer/iter"])
    par.broadcast_params_([tr.gs.flat]); tr.repack()
    if not str2bool(args.train):
        b = synth.make_batch(args.batchsize, h, w, seed=rank)
        pred, _, cams = tr.step(torch.from_numpy(b["ldr"]).to(dev), torch.from_numpy(b["sunpose_gt"]).to(dev), update=False)
        print("inference: cmf max %.4g, cam maxima %s" % (float(pred.max()), [round(float(c.max()), 4) for c in cams]))
        return 0
    for epoch in range(epoch0 + 1, args.epochs + 1):
        t0, acc = time.perf_counter(), 0.0
        for it in range(args.steps_per_epoch):
            seed = (epoch * 100003 + it) * world + rank
            if args.host_synth:
                b = synth.make_batch(args.batchsize, h, w, seed=seed)
                ldr, gt = torch.from_numpy(b["ldr"]).to(dev), torch.from_numpy(b["sunpose_gt"]).to(dev)
            else:
                b = synth.make_batch_device(args.batchsize, h, w, seed=seed, device=dev, crf=crf_train)
                ldr, gt = b["ldr"], b["sunpose_gt"]
            tr.step(ldr, gt, update=False, want_cams=False)
            par.allreduce_sum_([tr.gs.grad])
            tr.apply_gradients(gscale=1.0 / world)
            acc += tr.loss_dict()["sun_loss"] / args.steps_per_epoch
        if rank == 0:
            print("[epoch %d] train_loss_SUN %.6f  (%.2f s)" % (epoch, acc, time.perf_counter() - t0), flush=True)
            if epoch % 10 == 0:
                out = ckpt.sun_tensors(tr)
                out.update({"optimizer/m": tr.adam_m.cpu().numpy(), "optimizer/v": tr.adam_v.cpu().numpy(),
                            "optimizer/iter": tr.steps_done})
                print("Saved checkpoint for epoch %d: %s" % (epoch, mgr.save(out, epoch)))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
