#!/usr/bin/env python3
"""Helpers for tools/gpu_pmc_binding.sh.

  pmc_summarise.py <counter_collection.csv>          per-kernel, per-launch averages of every counter (one line each)
  pmc_summarise.py --json <passes.txt> <kernel_stats.csv> [<bench stdout of the traced run>]
        -> the JSON bench.py reads (profiles/rNN_pmc_binding_<workload>_<route>.json).  Every record carries a
        `_provenance` block: hash of the kernel sources it was collected on (splitp_amd._lib.source_hash), the git commit
        (SPLITP_GIT_COMMIT, exported by tools/gpu.sh - .git does not travel to the GPU box), sha16 of bench.py, the
        profiled command and the workload's shape (alignments x splits per launch, patterns).  bench.py prints a
        counter-based fraction only when hash and shape match the run it is describing.
"""
import collections
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PHASE_OF = {"k_sparse_score": "sparse", "k_sparse_slow": "chain", "k_gram_i8": "gram", "k_gram_i8_big": "gram", "k_eig_gv": "eigen",
            "k_eig_rr": "eigen_rr", "k_zero_i8": "zero", "k_scatter_i8": "scatter", "k_eig_init": "eigen_init", "k_reindex": "reindex",
            "k_subscore_pair": "subscore", "k_subscore_tri": "subscore", "k_subscore": "subscore_jacobi", "k_moments": "moment", "k_moments_mfma": "moment", "k_moments_reduce64": "moment_reduce", "k_enumerate_splits": "enumerate",
            "k_eig_cv": "eigen", "k_eig_ctv": "eigen_ctv", "k_eig4": "eig4", "k_fin_apply": "direct_apply", "k_fin_vec": "direct_vec",
            "k_fin_sturm": "direct_sturm"}


def base_name(kernel):
    return kernel.split("(")[0].replace("void ", "").split("<")[0].strip()


def per_kernel(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    with open(path) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k].add(row["Dispatch_Id"])
    for k in agg:
        n = len(calls[k])
        print(json.dumps({"kernel": k, "launches": n, "per_launch": {c: v / n for c, v in agg[k].items()}}))


def provenance(bench_stdout):
    sys.path.insert(0, ROOT)
    from splitp_amd import _lib

    prov = {"source_hash": _lib.source_hash(), "git_commit": os.environ.get("SPLITP_GIT_COMMIT", "unknown"),
            "bench_py_sha16": hashlib.sha256(open(os.path.join(ROOT, "bench.py"), "rb").read()).hexdigest()[:16],
            "command": os.environ.get("PMC_CMD", "unknown")}
    if bench_stdout and os.path.exists(bench_stdout):
        for line in open(bench_stdout):
            line = line.strip()
            if line.startswith("{") and '"metric"' in line:
                try:
                    d = json.loads(line)
                except ValueError:
                    continue
                cfg = d.get("config", {})
                prov["workload"] = d.get("metric")
                prov["shape"] = {k: cfg.get(k) for k in ("route", "alignments_per_rank_per_step", "splits_this_rank", "patterns")}
                prov["shape"]["workload_key"] = cfg.get("workload_key")
    return prov


def to_json(passes, stats, bench_stdout=None):
    out = {}
    for line in open(passes):
        line = line.strip()
        if not line.startswith("{"):
            continue
        rec = json.loads(line)
        phase = PHASE_OF.get(base_name(rec["kernel"]))
        if phase is None:
            continue
        d = out.setdefault(phase, {"kernel": rec["kernel"], "launches_profiled": rec["launches"]})
        d.update(rec["per_launch"])
    for phase, d in out.items():
        if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
            # MI355X_MICROARCH.md (HBM): FETCH_SIZE counts 64 B per 128-B request of wide coalesced reads -> x2; WRITE_SIZE
            # is exact for 16-B-per-lane stores; both are reported in KB by rocprofv3
            d["hbm_bytes_per_launch"] = 1024.0 * (2.0 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0))
        if "GRBM_GUI_ACTIVE" in d:
            d["kernel_cycles_from_GRBM_GUI_ACTIVE_div_8"] = d["GRBM_GUI_ACTIVE"] / 8.0   # rocprofv3 sums the 8 XCDs
    try:
        with open(stats) as fh:
            for row in csv.DictReader(fh):
                phase = PHASE_OF.get(base_name(row["Name"]))
                if phase in out:
                    out[phase]["trace_avg_ns"] = float(row["AverageNs"])
                    out[phase]["trace_calls"] = int(row["Calls"])
    except Exception as exc:   # noqa: BLE001
        out["_stats_error"] = str(exc)
    out["_provenance"] = provenance(bench_stdout)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "--json":
        to_json(*sys.argv[2:5])
    else:
        per_kernel(sys.argv[1])
