"""`python bench.py --gpus N` must bring N ranks up by itself (the flag was parsed and ignored, `--gpus 8` ran one rank
and printed n_gpus 1). The launch path is checked without a GPU through --dry-launch (gloo); on a box with fewer GPUs than asked
for, the real run has to refuse with a non-zero status instead of reporting a world size that did not run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_gpus_2_dry_launch_brings_two_ranks_up():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_launch"] is True and d["n_gpus"] == 2
    assert "rccl_ranks" in d and d["rccl_ranks"] is None       # gloo here: no communicator of the library's own to ask
    assert sorted(d["local_ranks"]) == [0, 1]          # two processes, distinct LOCAL_RANKs
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node=2" in r.stderr


def test_more_gpus_than_the_box_has_is_refused():
    import torch

    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    n = max(2, have + 1)
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode != 0
    assert f"--gpus {n}" in r.stderr and "refusing" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]      # no JSON line for a run that did not happen


def test_world_size_must_agree_with_gpus():
    env = _env()
    env.update({"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
