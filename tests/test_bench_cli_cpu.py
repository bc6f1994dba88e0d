"""bench.py's rank plumbing on a box without GPUs: `--gpus N` must never silently time fewer devices."""
import os
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HDRSKY_BENCH_ONE_CARD")}
    e.update(env)
    return subprocess.run([sys.executable, BENCH] + args, env=e, cwd=ROOT, capture_output=True, text=True, timeout=300)


def test_more_gpus_than_devices_is_refused():
    """`python bench.py --gpus 8` on a box with fewer devices exits non-zero before anything is timed (no JSON line)."""
    import torch
    n = torch.cuda.device_count()
    r = _run(["--gpus", str(n + 2), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "refusing" in r.stderr and '"metric"' not in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_world_size_must_equal_the_gpus_flag():
    """Inside a torchrun job of another size than --gpus says, every rank stops (before touching the GPU)."""
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout) and '"metric"' not in r.stdout
    r = _run(["--gpus", "0"])
    assert r.returncode != 0
