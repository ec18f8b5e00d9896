#!/usr/bin/env python3
"""Benchmark: splits scored/sec on the 10-taxon 100k-bp JC alignment (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1 without a launcher: this script starts its own N ranks - `python -m torch.distributed.run --nnodes=1
     --nproc-per-node N ... bench.py ...` as a child process, before torch or HIP are touched - and exits with the
     child's status; under a launcher (RANK / WORLD_SIZE set) it is one rank.)

One step = one pass of the hot path over one batch.  Default workload (`--workload config2`, BASELINE configs[1], the
configuration the metric is quoted on): score ALL 501 candidate splits of this rank's resident 10-taxon 100 k-bp
alignment - flattening + fp64 split score, the whole sparse route including its device-side hand-back chain - then
(N > 1) all-gather every rank's scores + status over RCCL, and copy them to pinned host memory.  Steps are issued
round-robin on `--lanes` lanes: each lane is its own library context (sp_ctx) bound to its own HIP stream with its own
work memory; the split plan and the alignments are shared read-only.  One host call (sp_score_plan_steps) enqueues
`--group` consecutive steps of a lane, each a complete pass into its own output slot; a lane is reused only after those
steps' scores are on the host and their status words were checked.  All K steps complete inside the timed region
(barrier + device sync on both sides); value = (splits scored by all ranks) / (max over ranks).

Partitions (`--shard`): `alignments` (default; weak scaling: every rank scores its own alignment) or `splits` (the
north-star partition, SURVEY 8e: ONE alignment replicated, its candidate-split set dealt to the ranks by cost class
with batch.shard_indices, one all-gather of the packed scores + status per group of steps; strong scaling).  At N > 1
the default invocation measures BOTH (`value` = the alignment-sharded whole-node rate on config 2, the split-sharded
runs on config 2 and config 4 beside it under `partitions`).
Other workloads: config5 (batch of 32 simulated 12-taxon alignments x 2035 splits, one device pass per step; the
batch is dealt to the ranks), config3 / config4 (16 / 20 taxa, 1 M bp, all splits, subflattening route: the kernels run in
the alignment's own context on one compute stream; two lanes there = two sets of result buffers and copy streams, so
that step i's scores and status words travel to the host beside the kernel of step i + 1).
`--mode dropin` additionally times the literal README loop (README.md:37-41) on the drop-in functions.

Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` for the dominant kernel - counter-based
fractions only from a PMC record under profiles/ whose source hash and workload shape match this run -,
`north_star_pipeline` (30 steps of the dense route: histogram table -> dense matrices -> MFMA Gram -> eigen, timed per
phase after the main region) and `cpu_baseline` (the oracle's faithful restatement of the reference CPU path on the
same table: all 501 splits with default BLAS threads and again with OMP_NUM_THREADS=1, both as child processes started
before any GPU call)."""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

# Lanes keep several steps in flight on separate HIP streams next to RCCL's own stream; with the default of 4 hardware
# queues two of those streams can share a queue and serialise, so ask the HIP runtime for 8 (read at HIP
# initialisation, hence before torch is imported).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this pool's host driver

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BRANCH = 0.05
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_MFMA_PEAK_TF = 78.6     # AMD MI355X FP64 matrix peak; the local guide has no f64 row (77.7 measured)
INT8_MFMA_PEAK_TOPS = 5033.0  # dense int8: 256 CUs x 4 SIMDs x 65536 ops per v_mfma_i32_32x32x32_i8 / 32 cycles x 2.4 GHz (guide: 2x the bf16 rate)
N_CU = 256
CLOCK_GHZ = 2.4              # MI355X_MICROARCH.md: max clock
LDS_GATHER_PEAK_TBS = N_CU * 128 * CLOCK_GHZ / 1e3   # MI355X_MICROARCH.md (LDS): ds_read2_b64 = 128 B/clk/CU -> 78.6 TB/s
WORKLOADS = {
    #          taxa  sites      alignments  method
    "config2": (10, 100_000, 1, "flattening"),
    "config3": (16, 1_000_000, 1, "subflattening"),
    "config4": (20, 1_000_000, 1, "subflattening"),
    "config5": (12, 100_000, 32, "flattening"),
}
WL_TEXT = {
    "config2": "BASELINE configs[1]: 10-taxon balanced tree (branch 0.05, JC), 100k bp, all 501 splits, "
               "flattening + fp64 split score (whole sparse route incl. its device-side hand-back chain), scores "
               "+ status copied to the host every step",
    "config3": "BASELINE configs[2]: 16-taxon balanced tree, 1M bp, all 32751 splits, subflattening route, fp64",
    "config4": "BASELINE configs[3]: 20-taxon balanced tree, 1M bp, all 524267 splits, subflattening route, fp64",
    "config5": "BASELINE configs[4]: batch of 32 device-simulated 12-taxon alignments (100k bp, branch 0.05, JC) x "
               "all 2035 splits in ONE device pass per step (sparse route + device-side chain); fp64 eigen "
               "arithmetic - stricter than the config's fp32 wording (decision: DESIGN.md section 7)",
}


# ----------------------------------------------------------------------------------------------- CPU baseline
def _cpu_leg_main(argv):
    """Child process: the oracle's loops layer (port of constructions.py:31-55 + phylogenetics.py:280-300) over the
    splits of the benchmark table, in all_splits order, until done or out of budget.  Prints one JSON line."""
    n_taxa, n_sites, budget = int(argv[0]), int(argv[1]), float(argv[2])
    import numpy as np  # noqa: F401
    from oracle import splitp_oracle as O
    from splitp_amd import synthetic as syn

    names = syn.taxa_names(n_taxa)
    sites = syn.simulate_sites(n_taxa, n_sites, BRANCH, seed=1)
    keys, counts = syn.pattern_table(sites)
    table = syn.table_as_dict(keys, counts, n_taxa, total=n_sites)
    splits = list(O.all_splits(names))
    # stratified order (every 21st split first, ...): a run cut short by the budget still covers every size class
    order = [i for off in range(21) for i in range(off, len(splits), 21)]
    t0 = time.perf_counter()
    done = 0
    sample = {}
    for i in order:
        sc = O.split_score(O.flattening(splits[i], table, "reduced"))
        if i in (0, 250, 500):
            sample[i] = float(sc)
        done += 1
        if time.perf_counter() - t0 > budget:
            break
    dt = time.perf_counter() - t0
    try:
        from threadpoolctl import threadpool_info

        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    print(json.dumps({"done": done, "total": len(splits), "seconds": dt, "threads": int(threads), "scores": sample}))


def start_cpu_legs(n_taxa, n_sites, budget):
    """Both legs start NOW (before any GPU call of this process) and run side by side: default BLAS threads, and
    OMP_NUM_THREADS=1."""
    legs = {}
    for name, env_extra in (("default_threads", {}), ("one_thread", {"OMP_NUM_THREADS": "1", "OPENBLAS_NUM_THREADS": "1",
                                                                        "MKL_NUM_THREADS": "1"})):
        env = dict(os.environ)
        env.update(env_extra)
        legs[name] = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--_cpu-leg", str(n_taxa), str(n_sites),
                                       str(budget)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True)
    return legs


def collect_cpu_legs(legs):
    res = {}
    for name, proc in legs.items():
        out, err = proc.communicate()
        if proc.returncode != 0:
            raise SystemExit(f"cpu baseline leg {name} failed:\n{err[-2000:]}")
        res[name] = json.loads(out.strip().splitlines()[-1])
    d, o = res["default_threads"], res["one_thread"]

    def what(r):
        return ("all %d splits" % r["total"]) if r["done"] == r["total"] else \
            ("%d of the %d splits (every 21st first: all size classes)" % (r["done"], r["total"]))
    return {"value": d["done"] / d["seconds"], "unit": "splits/s", "cores": d["threads"], "kind": "port",
            "sample": f"{what(d)}, FlatFormat.reduced + dense SVD (oracle loops layer = port of constructions.py:31-55 + "
                      f"phylogenetics.py:280-300), {d['seconds']:.1f} s, default BLAS threads ({d['threads']}); host has "
                      f"{os.cpu_count()} logical CPUs; both legs ran side by side as child processes before any GPU call",
            "one_thread": {"value": o["done"] / o["seconds"], "unit": "splits/s", "cores": 1,
                           "sample": f"{what(o)}, OMP_NUM_THREADS=1, {o['seconds']:.1f} s"},
            "os_cpu_count": os.cpu_count(), "scores": d["scores"]}


# ----------------------------------------------------------------------------------------------- rank launcher
def spawn_ranks(args):
    """--gpus N > 1 without a launcher: start N fresh ranks as a child process; this parent never touches torch/HIP."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.call(cmd)
    if rc != 0:
        print(f"bench.py: the {args.gpus}-rank child run failed with status {rc}", file=sys.stderr)
    sys.exit(rc)


# ----------------------------------------------------------------------------------------------- PMC records
def load_pmc(workload, route, shape):
    """The newest profiles/r*_pmc_binding_<workload>_<route>.json whose provenance matches THIS run: same kernel sources
    (splitp_amd._lib.source_hash) and the same work per launch (alignments x splits, patterns).  -> (record, file, None) or
    (None, None, reason).  A counter profile of another kernel build or another workload must not be divided by this run's
    launch time (VERDICT r2 / ADVICE r2)."""
    from splitp_amd import _lib

    want_hash = _lib.source_hash()
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_binding_{workload}_{route}.json")))
    if not paths:
        return None, None, f"no PMC record profiles/r*_pmc_binding_{workload}_{route}.json is committed"
    why = []
    for path in reversed(paths):
        try:
            rec = json.load(open(path))
        except Exception as exc:   # noqa: BLE001
            why.append(f"{os.path.basename(path)}: unreadable ({exc})")
            continue
        prov = rec.get("_provenance") or {}
        if prov.get("source_hash") != want_hash:
            why.append(f"{os.path.basename(path)}: collected on kernel sources {prov.get('source_hash')}, this tree is {want_hash}")
            continue
        ps = prov.get("shape") or {}
        bad = [k for k in ("alignments_per_rank_per_step", "splits_this_rank", "patterns") if ps.get(k) != shape.get(k)]
        if bad:
            why.append(f"{os.path.basename(path)}: profiled work per launch differs in {bad} ({ {k: ps.get(k) for k in bad} } "
                       f"there, { {k: shape.get(k) for k in bad} } here)")
            continue
        return rec, os.path.relpath(path, ROOT), None
    return None, None, "; ".join(why)


# ----------------------------------------------------------------------------------------------- roofline
def real_work(status, a_arr, n_taxa, patterns, gram_rows_max):
    """What one launch really computes (from the status words of the last step): sparse half products of the general
    path (the first one is a scatter of 4 rows and not counted) x non-zeros x 4 block columns, and the dense products of
    the Gram path (row sides of up to `gram_rows_max` ids iterate on the exact integer Gram matrix)."""
    import numpy as np

    if status is None:
        return None
    st = np.asarray(status).reshape(-1, len(a_arr))
    k = np.minimum(a_arr, n_taxa - a_arr)
    rows = 4.0 ** k
    general = rows > gram_rows_max
    halves = int(np.sum(np.maximum((st[:, general] >> 8) - 1, 0)))
    dense_products = (st[:, ~general] >> 8).astype(np.float64)
    dense_fma = float(np.sum(dense_products * (rows[~general] ** 2) * 4.0))
    return {"sparse_half_products": halves, "nnz": int(patterns), "block_columns": 4,
            "fma": halves * int(patterns) * 4 + int(dense_fma),
            "lds_gather_bytes": halves * int(patterns) * 4 * 8,
            "gram_path_items": int(np.count_nonzero(~general)) * st.shape[0], "their_dense_G_products": int(dense_products.sum()),
            "note": "general path: W = C^T V / Y = C W on the D non-zeros (one 8-byte LDS gather per non-zero and column); "
                    f"row sides of <= {gram_rows_max} ids iterate on the exact integer Gram matrix instead (R^2 x 4 FMA per product)"}


def lds_binding(pmc, sec, work):
    """LDS-array busy fraction (replays included) and the WORK-based fractions next to it."""
    avail = N_CU * sec * CLOCK_GHZ * 1e9
    lds = pmc["SQ_LDS_IDX_ACTIVE"]
    conf = pmc.get("SQ_LDS_BANK_CONFLICT", 0.0)
    valu_quad = pmc.get("SQ_ACTIVE_INST_VALU", 0.0)
    out = {"lds_array_cycles_per_launch": lds, "lds_array_cycles_available": avail, "lds_busy_frac": lds / avail,
           "bank_conflict_share": conf / lds if lds else None,
           "valu_issue_frac": valu_quad * 4.0 / (4 * avail) if valu_quad else None,
           "waves_parked_frac": (pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"]) if pmc.get("SQ_WAIT_ANY") and pmc.get("SQ_WAVE_CYCLES") else None,
           "real_work": work,
           "cycle_base": f"{N_CU} CUs x launch_ms x {CLOCK_GHZ} GHz"}
    useful = {"conflict_free_lds_frac": (lds - conf) / avail,
              "note": "work / peak, not busy time: LDS-array cycles without bank-conflict replays / cycles available; bytes the "
                      "sparse products gather from LDS / the ds_read2_b64 rate (128 B/clk/CU, MI355X_MICROARCH.md LDS table); "
                      "fp64 FMA executed / the fp64 peak"}
    if work:
        useful["lds_gather_TBs"] = work["lds_gather_bytes"] / sec / 1e12
        useful["lds_gather_frac_of_read_peak"] = work["lds_gather_bytes"] / sec / 1e12 / LDS_GATHER_PEAK_TBS
        useful["fp64_TFLOPs"] = 2.0 * work["fma"] / sec / 1e12
        useful["fp64_frac_of_peak"] = 2.0 * work["fma"] / sec / 1e12 / FP64_MFMA_PEAK_TF
    out["useful"] = useful
    return out


def roofline_block(m):
    """`roofline` of the dominant kernel of measurement `m` (dict returned by measure())."""
    import numpy as np

    dom, sec = m["dom"], m["dom_ms_alone"] * 1e-3
    n_taxa, n_sites, a_arr = m["n_taxa"], m["n_sites"], m["a_arr"]
    k_small = np.minimum(a_arr, n_taxa - a_arr).astype(np.float64)
    per_al = m["items_per_launch"] / max(len(a_arr), 1)            # alignments per launch
    algo_bytes = float(len(a_arr) * (4.0 * n_sites + 4.0 * 4.0 ** n_taxa)) * per_al       # 4L + 4*4^n per split
    algo_flops = float(np.sum(2.0 * 4.0 ** n_taxa * 4.0 ** k_small)) * per_al            # 2*4^n*4^k per split
    names = {"gram": "k_gram_i8_big<2,int> (int8-limb MFMA Gram, 128 x 128 tiles)", "eigen": "k_eig_gv / k_eig_rr (fp64 MFMA)",
             "sparse": "k_sparse_score (one workgroup per split: CSC/CSR lists + 4-wide block in LDS)",
             "scatter": "k_zero_i8 + k_scatter_i8", "reindex": "k_reindex", "subscore": "k_subscore_pair (two splits a wave; k_subscore_tri with subscore_pair = 0)",
             "chain": "k_sparse_slow (persistent workgroups: lists-in-global / all-global / 8-wide forms of the sparse kernel)"}
    roof = {"kernel": names.get(dom, dom), "launch_ms": m["dom_ms_alone"], "launch_ms_in_timed_region": m["dom_ms_region"],
            "launches_in_flight": m["in_flight"], "phase_ms_per_step": m["phases_per_step"], "traffic": None}
    pmc_all, pmc_file, pmc_why = load_pmc(m["workload"], m["route"], m["shape"])
    pmc = (pmc_all or {}).get(dom)
    roof["pmc_record"] = pmc_file if pmc else None
    if not pmc:
        roof["pmc_record_reason"] = pmc_why or f"the matching record {pmc_file} holds no counters of phase '{dom}'"
    if dom == "subscore":
        # SURVEY 8(d), subflattening path: per split a Gram 2 m^2 m' on the (3k+1) x (3(n-k)+1) block, Householder
        # tridiagonalisation 4/3 m^3 and Sturm multisection (17 passes x 32 shifts x m steps of the minor recurrence, 4
        # flops each: subtract, multiply, fused multiply-add; rounds 2 - 3: 13 passes x 64 shifts); bound: fp64 VALU issue
        mm = 3.0 * k_small + 1.0
        mp = 3.0 * (n_taxa - k_small) + 1.0
        flops = float(np.sum(2.0 * mm * mm * mp + 4.0 / 3.0 * mm ** 3 + 17 * 32 * 4.0 * mm)) * per_al
        roof.update({"bound": "fp64-valu", "achieved": flops / sec / 1e12, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                     "frac": flops / sec / 1e12 / FP64_MFMA_PEAK_TF, "algorithmic_flops_per_launch": flops,
                     "note": "two splits per wave, a lane per row; per-step scalar work (norms, reciprocals, reflector coefficients) "
                             "and the reductions are done by all 32 lanes of a half, so the useful-flop fraction is small by construction; what the "
                             "units do is in `binding` (vector issue, LDS, waves parked) (SURVEY 8d: 'fp64 VALU / launch "
                             "latency; report splits/s and achieved fp64 FLOP/s'); peak = the fp64 vector rate (= the fp64 "
                             "matrix rate on this part)"})
        if pmc:
            avail = N_CU * sec * CLOCK_GHZ * 1e9
            roof["traffic"] = pmc.get("hbm_bytes_per_launch")
            roof["binding"] = {
                "valu_issue_frac": pmc.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (4 * avail),
                "lds_busy_frac": pmc.get("SQ_LDS_IDX_ACTIVE", 0.0) / avail,
                "bank_conflict_share": (pmc.get("SQ_LDS_BANK_CONFLICT", 0.0) / pmc["SQ_LDS_IDX_ACTIVE"]) if pmc.get("SQ_LDS_IDX_ACTIVE") else None,
                "mfma_busy_frac": pmc.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4 * avail),
                "waves_parked_frac": (pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"]) if pmc.get("SQ_WAIT_ANY") and pmc.get("SQ_WAVE_CYCLES") else None,
                "waves_per_launch": pmc.get("SQ_WAVES"),
                "note": "VALU issue slots used / available (4 SIMDs per CU), LDS-array cycles / available, matrix-core busy "
                        "cycles / available, share of a wave's lifetime spent waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES)"}
        return roof
    survey = {"scatter_phase": {"bound": "hbm", "algorithmic_bytes_per_launch": algo_bytes,
                                "achieved": algo_bytes / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": algo_bytes / sec / 1e9 / HBM_PEAK_GBS},
              "gram_phase": {"bound": "mfma", "algorithmic_flops_per_launch": algo_flops,
                             "achieved": algo_flops / sec / 1e12, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                             "frac": algo_flops / sec / 1e12 / FP64_MFMA_PEAK_TF},
              "note": "SURVEY 8(d)'s per-unit figures (dense 4^a x 4^b flattening: 4L + 4*4^n bytes, 2*4^n*4^k Gram flops "
                      "per split) x the units one launch processes / the dominant kernel's own launch duration.  A "
                      "fraction above 1 means the route does not perform that dense work at all: it is reported because "
                      "the survey asks for it, it is NOT the bound the kernel is subject to (see `bound`)"}
    roof["survey_8d"] = survey
    if dom in ("gram", "eigen", "scatter", "reindex"):
        roof.update(dense_route_fractions(m["phases_per_step"], pmc_all, pmc_file, pmc_why))
        return roof
    if dom in ("sparse", "chain"):
        # What binds the sparse kernels is the LDS array and VALU issue, not HBM.  Binding fraction =
        # LDS-array cycles one launch used (rocprofv3 SQ_LDS_IDX_ACTIVE summed over the CUs, from the matching PMC record) /
        # LDS-array cycles available in the launch's own duration (CUs x duration x clock), the duration measured live here.
        roof.update({"bound": "lds", "peak": N_CU * CLOCK_GHZ * 1e9 / 1e12, "unit": "T LDS-array cycles/s"})
        if pmc:
            work = real_work(m["status"], a_arr, n_taxa, m["n_patterns"], 256 if dom == "chain" else 64)
            b = lds_binding(pmc, sec, work)
            roof.update({"achieved": pmc["SQ_LDS_IDX_ACTIVE"] / sec / 1e12, "frac": b["lds_busy_frac"], "binding": b,
                         "traffic": pmc.get("hbm_bytes_per_launch")})
            if dom == "chain":
                roof["traffic_note"] = ("FETCH_SIZE x 2 + WRITE_SIZE of the persistent workgroups' slabs (entry lists, descriptors): "
                                        "L2 <-> fabric traffic, most of it served by the 256 MB Infinity Cache rather than HBM")
        else:
            roof.update({"achieved": None, "frac": None})
        return roof
    roof.update({"bound": "hbm", "achieved": survey["scatter_phase"]["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": survey["scatter_phase"]["frac"]})
    return roof


def dense_route_fractions(phases_per_step, pmc_all, pmc_file, pmc_why):
    """Dense route (the north-star pipeline): its longest single kernel is the int8-limb MFMA Gram, reported on EXECUTED
    work - int8 operations one launch issues to the matrix cores (rocprofv3 SQ_INSTS_VALU_MFMA_MOPS_I8 x 512 from the
    matching PMC record: compact matrices, upper triangle, 2 x 2 limb products) / the Gram phase's duration measured live,
    against the dense int8 peak; the scatter phase the same way on the HBM bytes its two kernels move (PMC) against the HBM
    peak.  Without a matching record both fractions are null and say why."""
    out = {"kernel": "k_gram_i8_big<2,int> (int8-limb MFMA Gram, 128 x 128 tiles)", "launch_ms": phases_per_step.get("gram"),
           "bound": "mfma", "peak": INT8_MFMA_PEAK_TOPS, "unit": "TOP/s (int8)", "achieved": None, "frac": None,
           "pmc_record": pmc_file if pmc_all else None}
    if not pmc_all:
        out["pmc_record_reason"] = pmc_why
    g = (pmc_all or {}).get("gram") or {}
    gram_s = phases_per_step.get("gram", 0.0) * 1e-3
    ops = g.get("SQ_INSTS_VALU_MFMA_MOPS_I8", 0.0) * 512.0
    if ops and gram_s:
        out.update({"achieved": ops / gram_s / 1e12, "frac": ops / gram_s / 1e12 / INT8_MFMA_PEAK_TOPS,
                    "executed_int8_ops_per_launch": ops, "traffic": g.get("hbm_bytes_per_launch"),
                    "note": "executed work, not SURVEY 8(d)'s fp64 count of the uncompacted problem (kept under survey_8d); "
                            "peak = 256 CUs x 4 SIMDs x one v_mfma_i32_32x32x32_i8 (65536 ops) per 32 cycles x 2.4 GHz"})
        if g.get("SQ_VALU_MFMA_BUSY_CYCLES") and g.get("kernel_cycles_from_GRBM_GUI_ACTIVE_div_8"):
            out["mfma_busy_frac_pmc"] = g["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * N_CU * g["kernel_cycles_from_GRBM_GUI_ACTIVE_div_8"])
    zs = [(pmc_all or {}).get("zero") or {}, (pmc_all or {}).get("scatter") or {}]
    sc_bytes = sum(z.get("hbm_bytes_per_launch", 0.0) for z in zs)
    sc_s = phases_per_step.get("scatter", 0.0) * 1e-3
    out["scatter_phase"] = {"kernels": "k_zero_i8 + k_scatter_i8", "bound": "hbm", "ms": phases_per_step.get("scatter"),
                            "hbm_bytes_per_launch_pmc": sc_bytes or None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "achieved": sc_bytes / sc_s / 1e9 if sc_bytes and sc_s else None,
                            "frac": sc_bytes / sc_s / 1e9 / HBM_PEAK_GBS if sc_bytes and sc_s else None}
    # k_eig4 streams every split's int32 Gram matrix once per product, 3 products per split at this workload (certified
    # stop): 501 splits x R^2 x 4 bytes x 3 - bound by that stream (HBM / Infinity Cache), one workgroup per split
    g_bytes = (pmc_all or {}).get("eig4", {}).get("hbm_bytes_per_launch")
    eig_s = phases_per_step.get("eigen", 0.0) * 1e-3
    out["eigen_phase"] = {"ms": phases_per_step.get("eigen"), "kernel": "k_eig4<int>", "bound": "hbm",
                          "hbm_bytes_per_launch_pmc": g_bytes, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "achieved": g_bytes / eig_s / 1e9 if g_bytes and eig_s else None,
                          "frac": g_bytes / eig_s / 1e9 / HBM_PEAK_GBS if g_bytes and eig_s else None,
                          "note": "certified 4-wide block iteration on the exact Gram matrix, one 1024-thread workgroup per "
                                  "split streaming G (thread = row, V broadcast from LDS), MFMA 4 x 4 Gram + Cholesky-QR, "
                                  "stop on a certificate (gap + error bound); what it cannot certify goes to the direct "
                                  "solver (Householder + Sturm)"}
    return out


# ----------------------------------------------------------------------------------------------- one measurement
class Env:
    pass


def measure(env, workload, shard, route, steps, warmup, spinup, lanes_arg, group_arg, alignments_arg=0, timing=True):
    """K steps of `workload` under partition `shard` on route `route`, timed as the contract asks (barrier + device sync on
    both sides, max over ranks).  Returns a dict with everything the JSON line and the roofline block need."""
    import ctypes as C

    import numpy as np
    import torch

    import splitp_amd as sp
    from splitp_amd import _lib, batch
    from splitp_amd import simulation as sim
    from splitp_amd import synthetic as syn
    from splitp_amd.device import Context

    rank, world, dist, dev_t = env.rank, env.world, env.dist, env.dev_t
    n_taxa, n_sites, wl_aligns, method_name = WORKLOADS[workload]
    names = syn.taxa_names(n_taxa)
    shard_splits = shard == "splits"
    method = sp.Method.flattening if method_name == "flattening" else sp.Method.subflattening
    code = batch._method_code(method, route if method_name == "flattening" else "auto")
    use_plan = method_name == "flattening" and route in ("auto", "sparse")

    # ---- candidate splits: the full all_splits list, or this rank's shard of it -------------------------------
    taxa_all, a_all = sp.encode_all_splits(n_taxa)
    n_splits_total = len(a_all)
    # subflattening workloads (configs 3 / 4): the splits are enumerated on the device (sp_score_all_splits_shard) - no
    # split list crosses the boundary; a rank's shard = the combinations rank, rank + P, ... of every size class
    enum_on_device = method_name == "subflattening"
    if shard_splits and enum_on_device:
        shards, _ = batch.shard_layout(n_taxa, world)
        mine = shards[rank]
        per = max(len(s) for s in shards)
    elif shard_splits:
        shards = batch.shard_indices(batch.split_costs(a_all, n_taxa, code), world)
        mine = shards[rank]
        per = max(len(s) for s in shards)
    else:
        mine = np.arange(n_splits_total)
        per = n_splits_total
    taxa_arr, a_arr = np.ascontiguousarray(taxa_all[mine]), np.ascontiguousarray(a_all[mine])
    n_mine = len(a_arr)

    # ---- synthetic input: this rank's alignments, resident in HBM before the timed region ------------------------
    if alignments_arg > 0:
        n_al_rank, seeds = alignments_arg, [1 + rank * alignments_arg + a for a in range(alignments_arg)]
    elif wl_aligns > 1:                     # a fixed batch dealt to the ranks (config 5)
        seeds = [1 + a for a in range(wl_aligns)][rank::world]
        n_al_rank = len(seeds)
    else:
        n_al_rank, seeds = 1, [1 if shard_splits else 1 + rank]
    aligns, tables = [], []
    tree = syn.balanced_tree(n_taxa)
    for seed in seeds:
        if n_taxa <= 10:
            sites = syn.simulate_sites(n_taxa, n_sites, BRANCH, seed=seed)
            keys, counts = syn.pattern_table(sites)
            aligns.append(sp.DeviceAlignment.from_arrays(keys, None, n_taxa, counts=counts, n_sites=n_sites, taxa=names))
            tables.append((keys, counts))
        else:                               # larger tables are born on the device (simulator + histogram kernels)
            d = sim.generate_device_alignment(tree, sim.JukesCantor(), n_sites, seed=seed, branch_length=BRANCH)
            d.taxa = tuple(names)
            aligns.append(d)
            tables.append(None)
    ctx0 = aligns[0].ctx
    n_patterns = int(len(aligns[0]))
    items_rank = n_al_rank * n_mine                     # (alignment, split) pairs this rank scores per step
    width_al = batch.packed_width(per)                  # doubles per alignment in the exchange buffer
    width = n_al_rank * width_al

    # ---- lanes: one library context + HIP stream + buffers each; `group` steps per host call ----------------------
    # (more lanes with RCCL: the all-gather adds latency to every group, not work)
    n_lanes = lanes_arg if lanes_arg > 0 else ((5 if dist is not None else 3) if use_plan and workload == "config2" else 1)
    if not use_plan:
        # The dense / subflattening routes run in the alignment's own context - ONE compute stream, kernels in order.  Two
        # lanes there are two sets of result buffers and two copy streams: the copy of step i's scores and status words
        # to the host (6.3 MB at config 4: 0.35 ms next to a 3.9 ms kernel) runs beside the kernel of step i + 1.
        n_lanes = lanes_arg if lanes_arg > 0 else 2
    compute_stream = torch.cuda.Stream(device=dev_t) if not use_plan else None
    # (steps per host call: at one rank the GPU step - 0.1 ms - hides the 25 us of host work per step and deeper queues
    # measured 1 % slower, 0.1019 against 0.1011 ms; next to RCCL four steps share one all-gather and one host call)
    group = group_arg if group_arg > 0 else (4 if use_plan and workload == "config2" and dist is not None else 1)
    if not use_plan:
        group = 1
    plan = batch.SplitPlan(ctx0, taxa_arr, a_arr, n_taxa) if use_plan else None
    lib = ctx0._lib
    taxa_p, a_p = _lib._ptr(taxa_arr, C.c_int32), _lib._ptr(a_arr, C.c_int32)
    al_handles = (C.c_void_p * len(aligns))(*[a.handle.value for a in aligns])
    one_handle = [(C.c_void_p * 1)(a.handle.value) for a in aligns]
    step_bytes = width * 8

    class Lane:
        def __init__(self):
            self.stream = torch.cuda.Stream(device=dev_t)
            self.ctx = Context(ctx0.device, stream=self.stream.cuda_stream) if use_plan else ctx0
            # per step and alignment: `per` scores (f64) then `per` status words (int32, padded) - batch.gather_scores' layout
            self.send = torch.zeros(group * width, dtype=torch.float64, device=dev_t)
            self.recv = torch.zeros(world * group * width, dtype=torch.float64, device=dev_t) if dist is not None else None
            self.host = torch.zeros(world * group * width, dtype=torch.float64).pin_memory()
            self.host_np = self.host.numpy()
            self.done = torch.cuda.Event()
            self.kdone = torch.cuda.Event()   # (shared compute stream: the lane's kernels are done, its copy may start)
            self.busy = 0            # steps in flight on this lane
            base = self.send.data_ptr()
            self.packed = n_al_rank == 1 or not use_plan
            if self.packed:
                self.sc_p = [base + a * width_al * 8 for a in range(n_al_rank)]
                self.st_p = [base + a * width_al * 8 + per * 8 for a in range(n_al_rank)]
            else:
                # one call scores all alignments: its outputs are alignment-major [n_al][n_mine] scores and status,
                # laid out as two blocks (scores of all alignments, then status of all alignments)
                self.sc_p = [base]
                self.st_p = [base + n_al_rank * n_mine * 8]

    lanes = [Lane() for _ in range(n_lanes)]
    host_s = [0.0]

    def launch(lane, g):
        t_h = time.perf_counter()
        with torch.cuda.stream(lane.stream if use_plan else compute_stream):
            if use_plan:
                if lane.packed:
                    for a in range(n_al_rank):
                        _lib.check(lib.sp_score_plan_steps(lane.ctx.handle, one_handle[a], 1, plan.handle, g,
                                                           C.c_void_p(lane.sc_p[a]), step_bytes, C.c_void_p(lane.st_p[a]), step_bytes))
                else:
                    _lib.check(lib.sp_score_plan_steps(lane.ctx.handle, al_handles, n_al_rank, plan.handle, g,
                                                       C.c_void_p(lane.sc_p[0]), step_bytes, C.c_void_p(lane.st_p[0]), step_bytes))
            elif enum_on_device:
                ctx0.sync_stream_with_torch()
                n_got = C.c_int64()
                for a, al in enumerate(aligns):
                    _lib.check(lib.sp_score_all_splits_shard(al.handle, code, 0, 0, rank if shard_splits else 0,
                                                             world if shard_splits else 1, C.byref(n_got), None,
                                                             C.c_void_p(lane.sc_p[a]), None, C.c_void_p(lane.st_p[a])))
                    assert n_got.value == n_mine, (n_got.value, n_mine)
            else:
                ctx0.sync_stream_with_torch()
                for a, al in enumerate(aligns):
                    _lib.check(lib.sp_score_splits_async(al.handle, taxa_p, a_p, n_mine, code, C.c_void_p(lane.sc_p[a]),
                                                         C.c_void_p(lane.st_p[a])))
            if dist is not None:
                dist.all_gather_into_tensor(lane.recv, lane.send)      # one collective per group of steps
            if use_plan:
                lane.host.copy_(lane.recv if dist is not None else lane.send, non_blocking=True)
                lane.done.record()
            else:
                lane.kdone.record()
        if not use_plan:
            with torch.cuda.stream(lane.stream):                       # the lane's copy stream
                lane.stream.wait_event(lane.kdone)
                lane.host.copy_(lane.recv if dist is not None else lane.send, non_blocking=True)
                lane.done.record()
        lane.busy = g
        host_s[0] += time.perf_counter() - t_h

    def lane_results(lane, r=None, s=0):
        """(scores [n_al, n_mine], status [n_al, n_mine]) views of step s of rank r's block in the lane's host buffer."""
        r = rank if r is None else r
        blk = lane.host_np[r * group * width + s * width:r * group * width + (s + 1) * width]
        if lane.packed:
            rows = blk.reshape(n_al_rank, width_al)
            return rows[:, :n_mine], rows[:, per:].view(np.int32)[:, :n_mine]
        sc = blk[:n_al_rank * n_mine].reshape(n_al_rank, n_mine)
        st = blk[n_al_rank * n_mine:].view(np.int32)[:n_al_rank * n_mine].reshape(n_al_rank, n_mine)
        return sc, st

    unresolved = [0]
    finished = [0]
    trace = [None]      # SPLITP_BENCH_TRACE=1: host time of every retire inside the timed region (diagnostic)

    def retire(lane):
        """Wait until the lane's scores are on the host (the unit of work is complete) and check the status words of this
        rank's own items: the device chain leaves no bit 1 (not handled) behind; bit 0 (upper estimate) is counted."""
        if not lane.busy:
            return
        lane.done.synchronize()
        if trace[0] is not None:
            trace[0].append(time.perf_counter())
        for s in range(lane.busy):
            sc, st = lane_results(lane, s=s)
            if not use_plan and not enum_on_device and (st & 3).any():
                # dense route: a split its eigen kernel could not certify goes to the library's direct solver here, on
                # the host side of the pipeline (never on the benchmark tables: counted in `direct_finished`)
                for a, al in enumerate(aligns):
                    finished[0] += batch.finish_async(al, taxa_arr, a_arr, sc[a], st[a])
            if (st & 2).any():
                raise SystemExit("bench.py: a split came back unhandled (status bit 1) - the device chain must be complete")
            unresolved[0] += int(np.count_nonzero(st & 1))
        lane.busy = 0

    last = [None, 0]

    def run(n_steps):
        i, turn = 0, 0
        while i < n_steps:
            lane = lanes[turn % len(lanes)]
            g = min(group, n_steps - i)
            retire(lane)
            launch(lane, g)
            last[0], last[1] = lane, g
            i += g
            turn += 1
        for lane in lanes:
            retire(lane)

    # The phase timers' event pools (1024 hipEventCreate per lane context) are filled HERE, ahead of every untimed launch:
    # created where timing is switched on - between the warmup steps and the timed region - they leave the GPU idle for
    # milliseconds right before the clock starts, and a 20-step region (2 ms: the driver's command) then begins on a
    # device that has dropped out of its boost state.  SPLITP_BENCH_LATE_EVENTS=1 restores the old order (A/B diagnostic).
    late_events = os.environ.get("SPLITP_BENCH_LATE_EVENTS", "0") == "1"
    if timing and not late_events:
        for ctx_ in ({id(l.ctx): l.ctx for l in lanes}).values():
            ctx_.enable_timing(True)
            ctx_.enable_timing(False)

    # untimed spin-up: a fresh box needs a few hundred ms of load before clocks, queues and the RCCL channel settle;
    # then the W warmup steps the contract asks for.  (The number of spin-up batches is agreed between the ranks.)
    t_spin = time.perf_counter()
    spin_batch = 48 if workload == "config2" else 2
    while spinup > 0:
        run(spin_batch)
        go = time.perf_counter() - t_spin < spinup
        if dist is not None:
            flag = torch.tensor([1 if go else 0], dtype=torch.int32, device=dev_t)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            go = bool(int(flag.item()))
        if not go:
            break
    run(warmup)
    for lane in lanes:
        lane.ctx.enable_timing(timing)
        lane.ctx.reset_timing()
    unresolved[0] = 0
    host_s[0] = 0.0
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    if os.environ.get("SPLITP_BENCH_TRACE", "0") == "1":
        trace[0] = []
    t0 = time.perf_counter()
    run(steps)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    retire_ms = [round((t - t0) * 1e3, 4) for t in trace[0]] if trace[0] is not None else None
    trace[0] = None
    host_us = host_s[0] / max(steps, 1) * 1e6
    # per-phase device time inside the timed region, summed over the lanes (HIP events on each lane's own stream)
    phases = {}
    for lane in ({id(l.ctx): l for l in lanes}).values():
        for k, (ms, n) in lane.ctx.phase_times().items():
            a = phases.setdefault(k, [0.0, 0])
            a[0] += ms
            a[1] += n
        lane.ctx.enable_timing(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev_t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    scores, status = (x.copy() for x in lane_results(last[0], s=last[1] - 1))

    # the dominant kernel's OWN duration: a few un-overlapped launches on one lane straight after the timed region
    # (inside the region the lanes overlap, so an event-bracketed launch also spans its neighbours' workgroups)
    ph = {k: v for k, v in phases.items() if v[1] > 0}
    dom, dom_ms_alone = None, None
    if ph:
        dom = max(ph, key=lambda k: ph[k][0])
        alone_n = max(3, min(50, steps))
        l0 = lanes[0]
        l0.ctx.enable_timing(True)
        l0.ctx.reset_timing()
        for _ in range(alone_n):
            launch(l0, 1)
            retire(l0)
        alone = l0.ctx.phase_times()
        l0.ctx.enable_timing(False)
        dom_ms_alone = alone[dom][0] / max(alone[dom][1], 1)

    total_items = steps * (n_al_rank * n_splits_total if shard_splits else world * items_rank)
    if not shard_splits and wl_aligns > 1 and alignments_arg == 0:
        total_items = steps * wl_aligns * n_splits_total          # the whole dealt batch
    if shard_splits and enum_on_device:
        par = (f"split-sharded x{world}: one alignment replicated, every rank enumerates and scores its share of every size "
               "class on the device (index mod P), all_gather of scores + status")
    elif shard_splits:
        par = f"split-sharded x{world}: one alignment replicated, candidate splits dealt by cost class, all_gather of scores + status"
    elif world > 1:
        par = f"alignment-sharded x{world}, all_gather of scores + status"
    else:
        par = "single GPU"
    launches_per_step = (ph[dom][1] / steps) if dom else 1.0
    return {
        "workload": workload, "route": route, "shard": shard, "n_taxa": n_taxa, "n_sites": n_sites, "a_arr": a_arr, "mine": mine,
        "elapsed": elapsed, "steps": steps, "value": total_items / elapsed, "ms_per_step": elapsed / steps * 1e3,
        "scaling": "strong" if (shard_splits or (wl_aligns > 1 and alignments_arg == 0)) else "weak",
        "host_us_per_step": host_us, "retire_ms": retire_ms, "group": group, "n_lanes": n_lanes, "n_al_rank": n_al_rank, "n_mine": n_mine,
        "n_splits_total": n_splits_total, "n_patterns": n_patterns, "parallelism": par, "unresolved": unresolved[0], "finished": finished[0],
        "scores": scores, "status": status, "tables": tables, "names": names,
        "dom": dom, "dom_ms_alone": dom_ms_alone,
        "dom_ms_region": (ph[dom][0] / ph[dom][1]) if dom else None,
        "in_flight": (ph[dom][0] / (elapsed * 1e3)) if dom else None,
        "phases_per_step": {k: round(v[0] / steps, 5) for k, v in ph.items()},
        "items_per_launch": items_rank / max(launches_per_step, 1e-9),
        "shape": {"alignments_per_rank_per_step": n_al_rank, "splits_this_rank": n_mine, "patterns": n_patterns},
    }


def config_block(m):
    return {"workload": WL_TEXT[m["workload"]], "workload_key": m["workload"], "route": m["route"],
            "alignments_per_rank_per_step": m["n_al_rank"], "lanes": m["n_lanes"], "steps_per_host_call": m["group"],
            "splits_per_alignment": m["n_splits_total"], "splits_this_rank": m["n_mine"], "patterns": m["n_patterns"],
            "parallelism": m["parallelism"], "unconverged_splits_in_timed_region": m["unresolved"],
            "splits_finished_by_the_direct_solver": m["finished"]}


def north_star_pipeline(env, main_scores):
    """BASELINE configs[1] says "dense flattening"; north_star says histogram -> 4^|A| x 4^|B| matrix -> MFMA Gram -> Jacobi /
    eigen.  The default (driver-run) invocation therefore also times 30 steps of `--route dense` on the same table after
    the main region and reports them per phase."""
    import numpy as np

    m = measure(env, "config2", "alignments", "dense", 30, 3, 0.0, 1, 1)
    pmc_all, pmc_file, pmc_why = load_pmc("config2", "dense", m["shape"])
    frac = dense_route_fractions(m["phases_per_step"], pmc_all, pmc_file, pmc_why)
    diff = float(np.max(np.abs(m["scores"][0] - main_scores[0]))) if main_scores is not None else None
    return {"what": "30 steps of the dense route on the same 10-taxon 100k-bp table, timed after the main region: k_reindex "
                    "(bitmaps + ranks per split) -> k_zero_i8 + k_scatter_i8 (compact int8-limb matrices) -> k_gram_i8_big "
                    "(int8 MFMA Gram, exact) -> k_eig4 (certified 4-wide block iteration, one workgroup per split streaming G); "
                    "all 501 splits per step",
            "value": m["value"], "unit": "splits/s", "ms_per_step": m["ms_per_step"], "steps": m["steps"],
            "phase_ms_per_step": m["phases_per_step"],
            "gram_phase": {k: frac.get(k) for k in ("kernel", "launch_ms", "bound", "achieved", "peak", "unit", "frac",
                                                    "executed_int8_ops_per_launch", "mfma_busy_frac_pmc", "traffic",
                                                    "pmc_record", "pmc_record_reason") if k in frac},
            "scatter_phase": frac["scatter_phase"], "eigen_phase": frac["eigen_phase"],
            "max_abs_score_diff_vs_sparse_route": diff}


# ----------------------------------------------------------------------------------------------- main
def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--_cpu-leg":
        return _cpu_leg_main(sys.argv[2:])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--shard", default="auto", choices=["auto", "alignments", "splits"],
                    help="auto = alignments; at --gpus > 1 on the default workload the split-sharded runs are measured beside it")
    ap.add_argument("--mode", default="batched", choices=["batched", "dropin"])
    ap.add_argument("--alignments", type=int, default=0, help="alignments per rank per step (0 = the workload's)")
    ap.add_argument("--lanes", type=int, default=0, help="steps in flight (one sp_ctx + HIP stream + buffers each); 0 = auto")
    ap.add_argument("--group", type=int, default=0, help="steps enqueued per host call (sp_score_plan_steps); 0 = auto")
    ap.add_argument("--spinup", type=float, default=1.0, help="seconds of untimed load before the warmup steps")
    ap.add_argument("--no-hipri", action="store_true", help="RCCL stream at normal priority (diagnostic)")
    ap.add_argument("--route", default="auto", choices=["auto", "dense", "sparse"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline-block", action="store_true",
                    help="skip the north_star_pipeline block (30 steps of the dense route after the timed region of the default run)")
    ap.add_argument("--cpu-budget", type=float, default=170.0, help="seconds per CPU leg (all 501 splits take ~60-120 s)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        spawn_ranks(args)                       # never returns
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    n_taxa, n_sites, _, _ = WORKLOADS[args.workload]
    default_run = args.workload == "config2" and args.route == "auto" and args.shard == "auto" and args.alignments == 0
    shard = "alignments" if args.shard == "auto" else args.shard
    # CPU baseline legs: children, started before any GPU call of this process (rank 0 at N = 1 only)
    legs = None
    if world == 1 and not args.no_cpu_baseline and args.workload == "config2":
        legs = start_cpu_legs(n_taxa, n_sites, args.cpu_budget)

    import numpy as np
    import torch

    n_dev = torch.cuda.device_count()
    if n_dev < max(world, 1) or local_rank >= n_dev:
        if legs:
            for p in legs.values():
                p.kill()
        raise SystemExit(f"bench.py: {world} rank(s), {n_dev} device(s) visible: one GPU per rank is required "
                         f"(--gpus {args.gpus})")
    dist = None
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:   # launched by torch.distributed.run (also with 1 rank)
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        # RCCL prints a version banner to STDOUT when its communicator is created; rank 0's stdout carries ONE JSON line,
        # so file descriptor 1 points at stderr until the communicator exists (forced here by a first barrier)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            try:
                opts = dist.ProcessGroupNCCL.Options()
                opts.is_high_priority_stream = not args.no_hipri   # the all-gather competes with queued scoring workgroups
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), pg_options=opts)
            except (TypeError, AttributeError):                     # older / newer torch without these options
                dist.init_process_group("nccl")
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    else:
        torch.cuda.set_device(0)

    import splitp_amd as sp
    from splitp_amd import synthetic as syn

    sp._lib.require_gpu()
    env = Env()
    env.rank, env.world, env.dist, env.dev_t = rank, world, dist, torch.device("cuda", torch.cuda.current_device())

    # CPU legs finish before the GPU is timed (they would compete for host cores with the launch loop)
    cpu = collect_cpu_legs(legs) if legs else None

    m = measure(env, args.workload, shard, args.route, args.steps, args.warmup, args.spinup, args.lanes, args.group,
                alignments_arg=args.alignments)
    extra = {}
    if default_run and not args.no_pipeline_block and world == 1:
        try:
            extra["north_star_pipeline"] = north_star_pipeline(env, m["scores"])
        except Exception as exc:   # noqa: BLE001 - the headline line must not be lost to a failure of this side block
            print(f"bench.py: north_star_pipeline block failed: {exc!r}", file=sys.stderr)
            extra["north_star_pipeline"] = {"error": repr(exc)}
    # (SPLITP_BENCH_PARTITIONS=1: take this branch at one rank too - how the one-GPU box rehearses it under torch.distributed.run)
    if default_run and (world > 1 or (dist is not None and os.environ.get("SPLITP_BENCH_PARTITIONS") == "1")):
        # the north-star partition beside the weak-scaling run, in the same invocation: ONE alignment, its
        # candidate-split set sharded over the ranks - on config 2 (63 splits per rank at 8 ranks: latency-bound) and on
        # config 4 (524 267 splits, subflattening route: where sharding the split set pays)
        parts = {"alignment_sharded_config2": {"value": m["value"], "unit": "splits/s", "ms_per_step": m["ms_per_step"],
                                               "scaling": "weak", "host_us_per_step": m["host_us_per_step"]}}
        for key, wl, st, wu in (("split_sharded_config2", "config2", max(args.steps, 20), args.warmup),
                                ("split_sharded_config4", "config4", max(4, min(args.steps, 10)), 2)):
            ms = measure(env, wl, "splits", "auto", st, wu, 0.2, args.lanes, args.group, timing=False)
            parts[key] = {"value": ms["value"], "unit": "splits/s", "ms_per_step": ms["ms_per_step"], "steps": st,
                          "scaling": "strong", "host_us_per_step": ms["host_us_per_step"], "parallelism": ms["parallelism"],
                          "splits_this_rank": ms["n_mine"], "workload": WL_TEXT[wl]}
        extra["partitions"] = parts

    if rank == 0:
        out = {
            "metric": "splits scored/sec (whole node), 10-taxon 100k-bp JC alignment" if args.workload == "config2"
                      else f"splits scored/sec (whole node), {args.workload}",
            "value": m["value"], "unit": "splits/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": m["ms_per_step"], "higher_is_better": True, "scaling": m["scaling"],
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "host_us_per_step": m["host_us_per_step"],
            # how many ranks the collective library really saw (1 = no process group: nothing was gathered)
            "rccl_ranks": int(dist.get_world_size()) if dist is not None else 1,
            "collective_backend": (str(dist.get_backend()) + " (RCCL over xGMI)") if dist is not None else None,
            "config": config_block(m),
            "roofline": roofline_block(m) if m["dom"] else None,
        }
        out.update(extra)
        if m.get("retire_ms") is not None:
            out["retire_ms"] = m["retire_ms"][:64]
        if cpu is not None:
            out["cpu_baseline"] = {k: v for k, v in cpu.items() if k != "scores"}
            # end-to-end sanity outside the timed region: the GPU scores of three splits equal the CPU leg's
            for i, ref in cpu["scores"].items():
                j = int(np.nonzero(m["mine"] == int(i))[0][0])
                assert abs(ref - m["scores"][0, j]) <= 1e-10, (i, ref, m["scores"][0, j])
        else:
            out["cpu_baseline"] = None
        if args.mode == "dropin" and world == 1 and args.workload == "config2":
            out["dropin"] = dropin_loop(sp, syn, m["tables"][0], m["names"], n_taxa, n_sites, m["scores"][0], cpu)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def dropin_loop(sp, syn, table_arrays, names, n_taxa, n_sites, batched_scores, cpu):
    """The literal README loop (README.md:37-41) on the drop-in functions with a plain dict, all 501 splits."""
    import numpy as np

    keys, counts = table_arrays
    table = syn.table_as_dict(keys, counts, n_taxa, total=n_sites)
    splits = list(sp.all_splits(names))
    t0 = time.perf_counter()
    got = []
    for split in splits:
        flat = sp.flattening(split, table, sp.FlatFormat.reduced)
        got.append(sp.split_score(flat))
    dt = time.perf_counter() - t0
    err = float(np.max(np.abs(np.array(got) - batched_scores)))
    assert err <= 1e-10, err
    res = {"value": len(splits) / dt, "unit": "splits/s", "seconds": dt,
           "what": "for split in all_splits: split_score(flattening(split, dict, FlatFormat.reduced)) - unchanged README "
                   "loop on splitp_amd's drop-in functions, plain dict input, every flattening returned as a host ndarray",
           "max_abs_diff_vs_batched": err}
    if cpu:
        res["vs_cpu_baseline"] = res["value"] / cpu["value"]
    return res


if __name__ == "__main__":
    main()
