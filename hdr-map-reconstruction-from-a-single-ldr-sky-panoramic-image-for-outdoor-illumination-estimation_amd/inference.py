"""`python -m <pkg>.inference --indir D --outdir O` - the reference's inference CLI (inference.py:121-156).

For every *.jpg in --indir (sorted): BGR uint8 / 255 -> generator graph (inference.py:81-115) -> `<name>.hdr`
(Radiance RGBE) in --outdir.  Unlike the reference the models are built and the SKY / SUN checkpoints restored once,
not once per image, and --outdir is created.  JPEG decoding uses Pillow (OpenCV is not a dependency)."""
import argparse
import glob
import os

import numpy as np
import torch

from . import checkpoint as ckpt
from . import engine as E
from . import hdr_io
from . import kernels as K
from . import params as P


def load_ldr(path):
    from PIL import Image
    rgb = np.asarray(Image.open(path).convert("RGB"), np.float32)
    return rgb[..., ::-1] / 255.0      # cv2.imread order (BGR), inference.py:142-145


def main(argv=None):
    cwd = os.getcwd()
    ap = argparse.ArgumentParser(description="inference a model")
    ap.add_argument("--indir", type=str, default="None")
    ap.add_argument("--outdir", type=str, default="inference_output")
    ap.add_argument("--sky", type=str, default=os.path.join(cwd, "checkpoints/SKY"))
    ap.add_argument("--sun", type=str, default=os.path.join(cwd, "checkpoints/SUN"))
    ap.add_argument("--distortion-aware", default="", metavar="PARTS",
                    help="the checkpoints were trained with these layer families as distortion_aware_ops layers: comma list of "
                         "res, sunpose, decoders, or all (train.py --distortion-aware)")
    args = ap.parse_args(argv)
    if args.indir == "None":
        raise SystemExit("Please specify your input LDR directory")
    files = sorted(glob.glob(os.path.join(args.indir, "*.jpg")))
    os.makedirs(args.outdir, exist_ok=True)
    nets, shape = None, None
    for f in files:
        ldr = load_ldr(f)
        h, w, _ = ldr.shape
        if nets is None or shape != (h, w):
            gen = P.init_params(P.generator_spec(h, w), 0)
            sun = P.init_params(P.sunpose_spec(h, w), 1)
            t, _ = ckpt.CheckpointManager(args.sky).restore()
            if t:
                print("Latest SKY checkpoint has restored!! (%d variables)" % ckpt.load_into(gen, t, "gen_model"))
            t, _ = ckpt.CheckpointManager(args.sun).restore()
            if t:
                print("Latest SUN checkpoint has restored!! (%d variables)" % ckpt.load_into(sun, t, "lin"))
            nets, shape = E.Nets(gen, sun, device="cuda:0", precise=False, im_height=h, im_width=w), (h, w)
        x = torch.from_numpy(np.ascontiguousarray(ldr[None])).to("cuda:0")
        pred = E.generator_forward(nets, x, compute=K.BF16, distortion_aware=args.distortion_aware)["y_final_lin"][0].cpu().numpy()
        name = os.path.split(f)[-1].split(".")[0] + ".hdr"
        hdr_io.write_hdr(os.path.join(args.outdir, name), pred)
        print("wrote", os.path.join(args.outdir, name))


if __name__ == "__main__":
    main()
