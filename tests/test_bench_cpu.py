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
    r = _run(["--gpus", "1", "--no-cpu-baseline"], env_extra={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"}, timeout=240)
    assert r.returncode != 0 and "--gpus 1 but WORLD_SIZE=2" in r.stderr


def test_pmc_records_are_hash_and_shape_checked(tmp_path, monkeypatch):
    """bench.load_pmc: a counter record is used only when it was collected on THESE kernel sources and for the same work per
    launch (VERDICT r2 / ADVICE r2: a constant read from a file must not describe another kernel or workload)."""
    import importlib
    import json

    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    from splitp_amd import _lib

    prof = tmp_path / "profiles"
    prof.mkdir()
    shape = {"alignments_per_rank_per_step": 1, "splits_this_rank": 501, "patterns": 8226}
    good = {"sparse": {"SQ_LDS_IDX_ACTIVE": 1.0}, "_provenance": {"source_hash": _lib.source_hash(), "shape": dict(shape)}}
    stale = {"sparse": {"SQ_LDS_IDX_ACTIVE": 2.0}, "_provenance": {"source_hash": "0123456789abcdef", "shape": dict(shape)}}
    other = {"sparse": {"SQ_LDS_IDX_ACTIVE": 3.0}, "_provenance": {"source_hash": _lib.source_hash(),
                                                                    "shape": dict(shape, alignments_per_rank_per_step=4)}}
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    rec, path, why = bench.load_pmc("config2", "auto", shape)
    assert rec is None and path is None and "no PMC record" in why
    (prof / "r01_pmc_binding_config2_auto.json").write_text(json.dumps(good))
    (prof / "r02_pmc_binding_config2_auto.json").write_text(json.dumps(stale))     # newer name, older sources
    (prof / "r03_pmc_binding_config2_auto.json").write_text(json.dumps(other))     # right sources, other workload shape
    rec, path, why = bench.load_pmc("config2", "auto", shape)
    assert why is None and path.endswith("r01_pmc_binding_config2_auto.json") and rec["sparse"]["SQ_LDS_IDX_ACTIVE"] == 1.0
    (prof / "r01_pmc_binding_config2_auto.json").unlink()
    rec, path, why = bench.load_pmc("config2", "auto", shape)
    assert rec is None and "0123456789abcdef" in why and "alignments_per_rank_per_step" in why
    # every committed record carries the provenance bench.py checks (whether it still matches the sources is decided at run
    # time: a stale record makes `frac` null with a reason, it does not break anything)
    monkeypatch.setattr(bench, "ROOT", ROOT)
    import glob
    mine = glob.glob(os.path.join(ROOT, "profiles", "r03_pmc_binding_*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r04_pmc_binding_*.json"))
    assert len(mine) >= 10      # five workloads a round
    for f in mine:
        prov = json.load(open(f))["_provenance"]
        assert len(prov["source_hash"]) == 16 and int(prov["source_hash"], 16) >= 0
        assert set(shape) <= set(prov["shape"]) and prov["command"].startswith("python3 bench.py")
