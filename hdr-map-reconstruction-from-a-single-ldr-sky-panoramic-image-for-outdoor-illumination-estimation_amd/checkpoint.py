"""Checkpoint surface of the reference (train.py:208-220,510-522; tf_utils.py:298-320; inference.py:60-79) in a native
format: same directories (`checkpoints/SKY`, `checkpoints/SUN`), same retention (`max_to_keep=5`), same cadence hook
(`ckpt.epoch % 10 == 0`), keys named after the reference's object-graph paths.  One `ckpt-<N>.npz` per save plus a
`checkpoint` text file naming the latest (the TF tensor-bundle format itself is SURVEY.md section 8f "next")."""
import glob
import os
import re

import numpy as np


class CheckpointManager:
    def __init__(self, directory, max_to_keep=5):
        self.directory, self.max_to_keep = directory, max_to_keep
        os.makedirs(directory, exist_ok=True)

    def _all(self):
        files = glob.glob(os.path.join(self.directory, "ckpt-*.npz"))
        return sorted(files, key=lambda f: int(re.search(r"ckpt-(\d+)\.npz$", f).group(1)))

    @property
    def latest_checkpoint(self):
        files = self._all()
        return files[-1] if files else None

    def save(self, tensors, epoch):
        files = self._all()
        n = int(re.search(r"ckpt-(\d+)\.npz$", files[-1]).group(1)) + 1 if files else 1
        path = os.path.join(self.directory, "ckpt-%d.npz" % n)
        np.savez(path, epoch=np.int64(epoch), **{k.replace("/", "|"): np.asarray(v) for k, v in tensors.items()})
        with open(os.path.join(self.directory, "checkpoint"), "w") as f:
            f.write('model_checkpoint_path: "ckpt-%d"\n' % n)
        for old in self._all()[:-self.max_to_keep]:
            os.remove(old)
        return path

    def restore(self, path=None):
        path = path or self.latest_checkpoint
        if path is None:
            return None, 0
        z = np.load(path)
        return {k.replace("|", "/"): z[k] for k in z.files if k != "epoch"}, int(z["epoch"])


def sky_tensors(trainer):
    """tf.train.Checkpoint(epoch, gen_model, dis_model, gen_optimizer, disc_optimizer) (train.py:208-213).  As in the
    reference, gen_model does NOT contain the sun-pose net; its optimizer slots do (they belong to optimizer_gen)."""
    out = {}
    for k, v in trainer.gs.w.items():
        if k.startswith("gen."):
            out["gen_model/" + k[4:].replace(".", "/")] = v.detach().cpu().numpy()
    for k, v in trainer.ds.w.items():
        out["dis_model/" + k[4:].replace(".", "/")] = v.detach().cpu().numpy()
    out["gen_optimizer/rms"] = trainer.gs.ms.detach().cpu().numpy()
    out["disc_optimizer/rms"] = trainer.ds.ms.detach().cpu().numpy()
    return out


def sun_tensors(trainer):
    """tf_utils.checkpoint_initialization: Checkpoint(epoch, lin=model, optimizer) (tf_utils.py:309-312)."""
    return {"lin/" + k[4:].replace(".", "/"): v.detach().cpu().numpy() for k, v in trainer.gs.w.items() if k.startswith("sun.")}


def load_into(params, tensors, prefix):
    """Copies `prefix/<path>` entries into an OrderedDict name->np.array (names use '.' separators)."""
    n = 0
    for k in list(params.keys()):
        key = prefix + "/" + k.replace(".", "/")
        if key in tensors:
            params[k] = np.asarray(tensors[key], np.float32)
            n += 1
    return n
