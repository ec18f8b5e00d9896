"""bench.py's rank handling, checked without a GPU: `--gpus N` without a launcher starts its own N ranks, and a rank
count that the visible devices cannot serve fails loudly instead of silently measuring one GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=ROOT)


@pytest.mark.timeout(300)
def test_gpus_flag_spawns_ranks_and_fails_loudly_without_devices():
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("this host has the devices: the loud-failure path cannot be provoked")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r.returncode != 0
    assert "2 rank(s)" in r.stderr and "device(s) visible" in r.stderr, r.stderr[-2000:]
    assert '"n_gpus"' not in r.stdout          # no benchmark line was printed


def test_world_size_must_match_gpus_flag():
    r = _run(["--gpus", "1", "--no-cpu-baseline"], env_extra={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"}, timeout=60)
    assert r.returncode != 0 and "--gpus 1 but WORLD_SIZE=2" in r.stderr
