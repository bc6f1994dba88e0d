"""Checkpoint surface of the reference (train.py:208-220,510-522; tf_utils.py:298-320; inference.py:60-79) in a native
format: same directories (`checkpoints/SKY`, `checkpoints/SUN`), same retention (`max_to_keep=5`), same cadence hook
(`ckpt.epoch % 10 == 0`), keys named after the reference's object-graph paths.  One `ckpt-<N>.npz` per save plus a
`checkpoint` text file naming the latest.  Interchange with the reference's own files:
a directory holding TF tensor-bundle checkpoints (`ckpt-N.index` + `ckpt-N.data-*`, written by `tf.train.Checkpoint`) is
restored through tf_bundle.py, and `export_tf_bundle` writes one."""
import glob
import os
import re

import numpy as np

_NUM = re.compile(r"ckpt-(\d+)\.(?:npz|index)$")


class CheckpointManager:
    def __init__(self, directory, max_to_keep=5):
        self.directory, self.max_to_keep = directory, max_to_keep
        os.makedirs(directory, exist_ok=True)

    def _all(self):
        files = glob.glob(os.path.join(self.directory, "ckpt-*.npz"))
        return sorted(files, key=lambda f: int(re.search(r"ckpt-(\d+)\.npz$", f).group(1)))

    @property
    def latest_checkpoint(self):
        """Newest `ckpt-N.npz` (native) or `ckpt-N.index` (TF tensor bundle) of the directory."""
        files = self._all() + glob.glob(os.path.join(self.directory, "ckpt-*.index"))
        files.sort(key=lambda f: (int(_NUM.search(f).group(1)), f.endswith(".npz")))
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
        if path.endswith(".index"):
            return import_tf_bundle(path[:-len(".index")])
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


# The last path component of a variable: the native files use the names the reference passes to add_weight ('w' / 'b',
# 'kernel_deconv2d' / 'bias_deconv2d': ops.py:30-37,76-108), whereas a tf.train.Checkpoint object graph is keyed by the
# ATTRIBUTE the layer stores the variable under (`self.w` / `self.biases`, `self.kernel` / `self.biases`: the same
# lines).  Both spellings are accepted when restoring.
_ALIASES = {"b": ("biases",), "kernel_deconv2d": ("kernel",), "bias_deconv2d": ("biases",)}


def _candidates(prefix, name):
    parts = name.split(".")
    yield prefix + "/" + "/".join(parts)
    for alt in _ALIASES.get(parts[-1], ()):
        yield prefix + "/" + "/".join(parts[:-1] + [alt])


def load_into(params, tensors, prefix, strict=True):
    """Copies `prefix/<path>` entries of a restored checkpoint into an OrderedDict name -> np.array (names use '.'
    separators).  Every loaded array must have the shape the model declares (a checkpoint written at another
    --imheight/--imwidth is rejected, not silently adopted).  strict: every variable of `params` must be found -
    a partial restore (e.g. a key-spelling mismatch that leaves the biases at their initial values) raises instead of
    producing a half-initialised model.  Returns the number of variables loaded."""
    n, missing = 0, []
    for k in list(params.keys()):
        key = next((c for c in _candidates(prefix, k) if c in tensors), None)
        if key is None:
            missing.append(k)
            continue
        v = np.asarray(tensors[key], np.float32)
        if tuple(v.shape) != tuple(np.shape(params[k])):
            raise ValueError("checkpoint variable %s has shape %s, the model expects %s for %s"
                             % (key, tuple(v.shape), tuple(np.shape(params[k])), k))
        params[k] = v
        n += 1
    if strict and missing:
        raise KeyError("checkpoint holds no `%s/...` entry for %d of %d variables (first: %s)"
                       % (prefix, len(missing), len(params), ", ".join(missing[:4])))
    return n


# ---- TF object-graph paths <-> native keys -------------------------------------------------------------------
# The reference's resLayer keeps its blocks in a python list attribute `sequence` (generator.py:41-44), which the
# object graph spells `res/sequence/<i>/...`; everything else already uses the attribute path.
def tf_path(key):
    return re.sub(r"^(gen_model/res)/(\d+)/", r"\1/sequence/\2/", key)


def native_path(key):
    return re.sub(r"^(gen_model/res)/sequence/(\d+)/", r"\1/\2/", key)


def import_tf_bundle(prefix):
    """(tensors, epoch) from a TF tensor-bundle checkpoint prefix (`.../ckpt-3`): variables keyed like the native format
    (`gen_model/...`, `dis_model/...`, `lin/...`); optimizer slots are not imported (RMSprop restarts its averages)."""
    from . import tf_bundle
    tensors = {native_path(k): v for k, v in tf_bundle.variable_tensors(tf_bundle.read_bundle(prefix)).items()}
    epoch = int(np.asarray(tensors.pop("epoch", 0)).reshape(-1)[0])
    tensors.pop("save_counter", None)
    return tensors, epoch


_ATTR = {"b": "biases", "kernel_deconv2d": "kernel", "bias_deconv2d": "biases"}     # add_weight name -> layer attribute (ops.py)


def _attr_path(key):
    """Native key -> the object-graph path a tf.train.Checkpoint of the reference's classes uses: `res/<i>` ->
    `res/sequence/<i>` and the LAST component spelled like the attribute the layer keeps the variable under."""
    parts = tf_path(key).split("/")
    if parts[0] in ("gen_model", "lin") and len(parts) >= 3 and parts[-1] in _ATTR and "sun" not in parts[1:2] \
            and not parts[-2].startswith("norm") and parts[-2] not in ("fc1", "fc2"):
        parts[-1] = _ATTR[parts[-1]]
    return "/".join(parts)


def export_tf_bundle(prefix, tensors, epoch, slots=None):
    """Writes the model variables of `tensors` (native keys; the flat optimizer buffers are skipped) as a TF tensor bundle
    in the layout of `tf.train.Checkpoint(epoch=..., gen_model=..., dis_model=..., gen_optimizer=..., disc_optimizer=...)
    .save` (train.py:208-220) / `Checkpoint(epoch, lin, optimizer)` (tf_utils.py:309-312): every variable under
    `<attribute path>/.ATTRIBUTES/VARIABLE_VALUE`, `epoch` and `save_counter`, per-variable optimizer slots
    (`slots`: {optimizer name: {native variable key: array}}, e.g. from `sky_slots`) under
    `<variable path>/.OPTIMIZER_SLOT/<optimizer>/rms/...`, and the `_CHECKPOINTABLE_OBJECT_GRAPH` string entry that
    `Checkpoint.restore` walks (tf_bundle.object_graph).  UNPINNED against TensorFlow itself: written from the public
    format definition; no TF build and no reference checkpoint exist in this environment to read it back with."""
    from . import tf_bundle
    out, names = {}, {}
    for k, v in tensors.items():
        if k.endswith("_optimizer/rms") or k.startswith("optimizer/"):
            continue
        path = _attr_path(k)
        out[path + tf_bundle.SUFFIX] = np.asarray(v)
        names[path] = k.split("/", 1)[-1]          # full_name: the variable's own name (informational in TF)
    for opt, per_var in (slots or {}).items():
        for k, v in per_var.items():
            out[_attr_path(k) + tf_bundle.SLOT + opt + "/rms" + tf_bundle.SUFFIX] = np.asarray(v, np.float32)
    out["epoch" + tf_bundle.SUFFIX] = np.asarray(epoch, np.int64)
    m = re.search(r"ckpt-(\d+)$", prefix)
    out["save_counter" + tf_bundle.SUFFIX] = np.asarray(int(m.group(1)) if m else 1, np.int64)
    out[tf_bundle.OBJECT_GRAPH_KEY] = tf_bundle.object_graph(list(out), names)
    tf_bundle.write_bundle(prefix, out)
    return prefix


def sky_slots(trainer):
    """Per-variable RMSprop slots of the SKY checkpoint's optimizers, cut out of the flat slot buffers:
    {"gen_optimizer": {native key: rms array}, "disc_optimizer": {...}} for the variables the checkpoint's object graph
    reaches (gen_model, dis_model; the sun-pose net's slots live in optimizer_gen but its variables are not in this graph)."""
    out = {"gen_optimizer": {}, "disc_optimizer": {}}
    for fp, opt, pre, model in ((trainer.gs, "gen_optimizer", "gen.", "gen_model/"), (trainer.ds, "disc_optimizer", "dis.", "dis_model/")):
        ms = fp.ms.detach().cpu().numpy()
        for k, (o, n, shape) in fp.offsets.items():
            if k.startswith(pre) and o < fp.ntrain:
                out[opt][model + k[len(pre):].replace(".", "/")] = ms[o:o + n].reshape(shape)
    return out
