"""`python -m <pkg>.train` - the reference's train.py CLI surface (train.py:527-545) on the MI355X step.

Flags keep the reference's names and defaults: --dir --lr(1e-4) --batchsize(32) --epochs(1000) --imheight(32)
--imwidth(128) --sky --sun --dorf --vgg.  There is no Laval dataset / dorfCurves.txt / vgg16.npy in this environment:
without --dir the loop trains on seeded synthetic batches (synth.make_batch); --vgg loads a real vgg16.npy when given.
Per epoch it prints the reference's scalar names (train.py:480-489), logs them to a TensorBoard event file
(<logdir>/tensorboard/SKY/<timestamp>/train, tb_logging.py) and every 10th epoch saves SKY / SUN checkpoints
with max_to_keep=5 (train.py:516-522).  Launch with torchrun for data parallelism (one process per GPU).
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
from . import tb_logging
from .trainer import Trainer


def fit_synthetic(tr, steps, batchsize, seed0=0, world=1, rank=0, crf=None, jpeg=True, exchange=None):
    """`steps` optimizer steps of the captured training step on seeded device-side synthetic batches (the epoch loop of
    `main` without logging / checkpoints): batch `it` is synth.make_batch_device(seed = (seed0 + it) * world + rank).
    Used by main(), by bench.py and the parity tests to produce TRAINED-LIKE weights from a committed procedure instead of
    a weight blob (there is no dataset and no published checkpoint)."""
    h, w, dev = tr.h, tr.w, tr.device
    bufs = getattr(tr, "_fit_bufs", None)
    for it in range(steps):
        b = synth.make_batch_device(batchsize, h, w, seed=(seed0 + it) * world + rank, device=dev, crf=crf, jpeg=jpeg)
        if bufs is None or bufs[0].shape[0] != batchsize:
            bufs = tr._fit_bufs = (b["ldr"].clone(), b["hdr_t"].clone(), b["sunpose_gt"].clone())
            tr.capture(*bufs)
        for dst, src in zip(bufs, (b["ldr"], b["hdr_t"], b["sunpose_gt"])):
            dst.copy_(src)
        tr.replay(hooks=exchange.hooks if exchange else None, pre_hooks=exchange.pre_hooks if exchange else None)
    return tr


def psnr_db(a, b, peak):
    """10 log10(peak^2 / mean((a-b)^2)) in float64."""
    a, b = a.double(), b.double()
    return float(10.0 * torch.log10(float(peak) ** 2 / ((a - b) ** 2).mean()))


def quality_report(tr, ldr, hdr_t, compute_modes=("BF16", "BF16X3")):
    """PSNR of the inference graph's y_final_gamma (the trainer's CURRENT weights, inference-mode BatchNorm) against the
    log-compressed target hdr_logCompression(hdr_t) (tf_utils.py:263-271), per compute mode, and between the modes."""
    from . import engine
    gen = {k[4:]: v for k, v in tr.gs.w.items() if k.startswith("gen.")}
    sun = {k[4:]: v for k, v in tr.gs.w.items() if k.startswith("sun.")}
    nets = engine.Nets(gen, sun, device=tr.device, precise=True, im_height=tr.h, im_width=tr.w)
    tgt = K.tonemap(hdr_t, False)
    peak = float(tgt.abs().max())
    ys = {m: engine.generator_forward(nets, ldr, compute=getattr(K, m))["y_final_gamma"] for m in compute_modes}
    rep = {"psnr_%s_vs_target_db" % m.lower(): round(psnr_db(y, tgt, peak), 3) for m, y in ys.items()}
    if len(compute_modes) == 2:
        a, b = (ys[m] for m in compute_modes)
        rep["psnr_%s_vs_%s_db" % tuple(m.lower() for m in compute_modes)] = round(psnr_db(a, b, float(b.abs().max())), 3)
    return rep


def main(argv=None):
    cwd = os.getcwd()
    ap = argparse.ArgumentParser(description="training the LDR->HDR sky model")
    ap.add_argument("--dir", type=str, default=None, help="dataset directory (TFRecords; not supported here -> synthetic)")
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--batchsize", type=int, default=32)
    ap.add_argument("--epochs", type=int, default=1000)
    ap.add_argument("--imheight", type=int, default=32)
    ap.add_argument("--imwidth", type=int, default=128)
    ap.add_argument("--sky", type=str, default=os.path.join(cwd, "checkpoints/SKY"))
    ap.add_argument("--sun", type=str, default=os.path.join(cwd, "checkpoints/SUN"))
    ap.add_argument("--dorf", type=str, default=None,
                    help="dorfCurves.txt (utils.getDoRF): camera response curves of the LDR synthesis; default: a 1/2.2 gamma")
    ap.add_argument("--vgg", type=str, default=None)
    ap.add_argument("--logdir", type=str, default=cwd,
                    help="TensorBoard scalars go to <logdir>/tensorboard/SKY/<timestamp>/train (tf_utils.py:282-292)")
    ap.add_argument("--no-tensorboard", action="store_true")
    ap.add_argument("--steps-per-epoch", type=int, default=8, help="synthetic mode: steps per epoch")
    ap.add_argument("--val-steps", type=int, default=0,
                    help="synthetic mode: validation batches per epoch through test_step (train.py:491-506); 0 = none")
    ap.add_argument("--no-graph", action="store_true", help="issue every launch eagerly instead of replaying hipGraphs")
    ap.add_argument("--dp-mode", default=None, choices=list(par.MODES),
                    help="gradient exchange between data-parallel replicas (parallel.GradientExchange)")
    ap.add_argument("--distortion-aware", default="", metavar="PARTS",
                    help="run these layer families as distortion_aware_ops layers (the variants the reference keeps commented "
                         "out): comma list of res (generator.py:14,18), sunpose (sunpose_net.py:11,16), decoders "
                         "(distortion_aware_ops.deconv2d in sky_decode / sun_decode), or all")
    ap.add_argument("--host-synth", action="store_true",
                    help="build the synthetic batches with numpy on the host (40 ms per batch of 32) instead of on the GPU")
    args = ap.parse_args(argv)

    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    rank, world, _ = par.init_from_env(device=dev)
    h, w = args.imheight, args.imwidth
    crf_train = crf_test = None
    if args.dorf:                    # utils.py:105-116, train.py:140-141: the first 175 curves train, the rest validate
        if args.host_synth:
            raise SystemExit("--dorf needs the device-side LDR synthesis (drop --host-synth)")
        crf_train, crf_test = synth.load_dorf(args.dorf)
        crf_test = crf_test if len(crf_test) else crf_train
    gen = P.init_params(P.generator_spec(h, w), 0)
    sun = P.init_params(P.sunpose_spec(h, w), 1)
    dis = P.init_params(P.discriminator_spec(), 2)
    vgg = P.load_vgg_npy(args.vgg) if args.vgg else P.init_params(P.vgg_spec(), 3)
    sky_mgr, sun_mgr = ckpt.CheckpointManager(args.sky), ckpt.CheckpointManager(args.sun)
    tensors, epoch0 = sky_mgr.restore()
    if tensors:
        n = ckpt.load_into(gen, tensors, "gen_model") + ckpt.load_into(dis, tensors, "dis_model")
        print("Latest SKY checkpoint has restored!! (%d variables)" % n)
    sun_t, _ = sun_mgr.restore()
    if sun_t:
        print("Latest SUN checkpoint has restored!! (%d variables)" % ckpt.load_into(sun, sun_t, "lin"))
    tr = Trainer(gen, sun, dis, vgg, device=dev, lr=args.lr, im_height=h, im_width=w, compute=K.BF16, world_size=world,
                 distortion_aware=args.distortion_aware)
    if tensors and "gen_optimizer/rms" in tensors:
        tr.gs.ms.copy_(torch.from_numpy(tensors["gen_optimizer/rms"])); tr.ds.ms.copy_(torch.from_numpy(tensors["disc_optimizer/rms"]))
    par.broadcast_params_([tr.gs.flat, tr.ds.flat]); tr.repack()
    ex = par.GradientExchange(tr, device=dev, mode=args.dp_mode)

    # The step is captured once (one hipGraph per segment, see trainer.py) on static input buffers that every batch is
    # copied into; losses are accumulated on the device and read back once per epoch.
    bufs, captured = None, False
    tb_train = tb_val = None
    if rank == 0 and not args.no_tensorboard:
        tb_train, tb_val, tb_dir = tb_logging.create_directories(args.logdir, "SKY")
    from .trainer import LOSS_SLOTS
    for epoch in range(epoch0 + 1, args.epochs + 1):
        t0 = time.perf_counter()
        loss_acc = torch.zeros(len(LOSS_SLOTS), dtype=torch.float32, device=dev)
        for it in range(args.steps_per_epoch):
            seed = (epoch * 100003 + it) * world + rank
            if args.host_synth:
                b = synth.make_batch(args.batchsize, h, w, seed=seed)
                ldr, hdr, gt = (torch.from_numpy(b[k]).to(dev) for k in ("ldr", "hdr_t", "sunpose_gt"))
            else:   # augmentation + target construction of train.py:42-94 on the GPU
                b = synth.make_batch_device(args.batchsize, h, w, seed=seed, device=dev, crf=crf_train)
                ldr, hdr, gt = b["ldr"], b["hdr_t"], b["sunpose_gt"]
            if args.no_graph:
                out = tr.step(ldr, hdr, gt, update=False)
                ex.reduce_all()
                tr.apply_gradients()
            else:
                if bufs is None:
                    bufs = (ldr.clone(), hdr.clone(), gt.clone())
                    out = tr.capture(*bufs)
                    captured = True
                for dst, src in zip(bufs, (ldr, hdr, gt)):
                    dst.copy_(src)
                tr.replay(hooks=ex.hooks, pre_hooks=ex.pre_hooks)   # gradients are scaled by 1/world in the optimizer
            loss_acc += tr.losses
        v = dict(zip(LOSS_SLOTS, (loss_acc / args.steps_per_epoch).tolist()))
        v["total_gen_loss"] = v["kl"] + 1000.0 * v["dog"] + v["adv"] + 10.0 * v["l1"] + 0.01 * v["perceptual"]
        v["total_disc_loss"] = 0.5 * (v["disc_generated"] + v["disc_real"])
        acc = v
        if epoch % 10 == 0 or args.val_steps > 0:
            # BatchNorm moving statistics are replica-local: what is saved / validated is rank 0's copy on every replica
            par.sync_moving_stats_(tr)
            if args.val_steps > 0:
                tr.refresh_eval()
        if rank == 0:
            names = (("gen_total_loss", "total_gen_loss"), ("gen_l1_loss", "l1"), ("gen_perceptual_loss", "perceptual"),
                     ("gen_DoG_loss", "dog"), ("gen_adv_loss", "adv"), ("gen_kl_div", "kl"),
                     ("disc_total_loss", "total_disc_loss"), ("disc_generated_loss", "disc_generated"), ("disc_real_loss", "disc_real"))
            if tb_train is not None:   # train.py:478-489, 513-514: one point per epoch
                tb_train.scalars({n: acc[k] for n, k in names}, step=epoch)
                tb_train.scalars({"g_out": float(out["gamma"].max()), "b_out": float(out["beta"].max())}, step=epoch)
                tb_train.flush()
            print("[epoch %d] %s  g_out=%.4f b_out=%.4f  Spends : %.2f(s)" %
                  (epoch, "  ".join("%s=%.5g" % (n, acc[k]) for n, k in names), float(out["gamma"].max()),
                   float(out["beta"].max()), time.perf_counter() - t0))
            if epoch % 10 == 0:
                print("Saved SKY checkpoint for epoch {}: {}".format(epoch, sky_mgr.save(ckpt.sky_tensors(tr), epoch)))
                print("Saved SUN checkpoint for epoch {}: {}".format(epoch, sun_mgr.save(ckpt.sun_tensors(tr), epoch)))
        if args.val_steps > 0:        # train.py:491-506: test_step over the validation split, same tags, `val` writer
            vacc = torch.zeros(len(LOSS_SLOTS), dtype=torch.float32, device=dev)
            for it in range(args.val_steps):
                vb = synth.make_batch_device(args.batchsize, h, w, seed=(7_000_003 + epoch * 1009 + it) * world + rank, device=dev,
                                              crf=crf_test)
                tr.test_step(vb["ldr"], vb["hdr_t"], vb["sunpose_gt"])
                vacc += tr.losses
            vv = dict(zip(LOSS_SLOTS, (vacc / args.val_steps).tolist()))
            vv["total_gen_loss"] = vv["kl"] + 1000.0 * vv["dog"] + vv["adv"] + 10.0 * vv["l1"] + 0.01 * vv["perceptual"]
            vv["total_disc_loss"] = 0.5 * (vv["disc_generated"] + vv["disc_real"])
            if rank == 0:
                if tb_val is not None:
                    tb_val.scalars({n: vv[k] for n, k in names}, step=epoch)
                    tb_val.flush()
                print("[epoch %d][val] %s" % (epoch, "  ".join("%s=%.5g" % (n, vv[k]) for n, k in names)))


if __name__ == "__main__":
    main()
