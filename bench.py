#!/usr/bin/env python3
"""Benchmark: splits scored/sec on the 10-taxon 100k-bp JC alignment (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1 without a launcher: this script starts its own N ranks - `python -m torch.distributed.run --nnodes=1
     --nproc-per-node N ... bench.py ...` as a child process, before torch or HIP are touched - and exits with the
     child's status; under a launcher (RANK / WORLD_SIZE set) it is one rank.)

One step = one pass of the hot path over one batch.  Default workload (`--workload config2`, BASELINE configs[1], the
configuration the metric is quoted on): score ALL 501 candidate splits of this rank's resident 10-taxon 100 k-bp
alignment - flattening + fp64 split score, the whole sparse route including its device-side hand-back chain, one
sp_score_plan_async call - then (N > 1) all-gather every rank's scores + status over RCCL, and copy them to pinned
host memory.  Steps are issued round-robin on `--lanes` lanes: each lane is its own library context (sp_ctx) bound to
its own HIP stream with its own work memory; the split plan and the alignments are shared read-only.  A lane is reused
only after its previous step's scores are on the host and its status words were checked.  All K steps complete inside
the timed region (barrier + device sync on both sides); value = (splits scored by all ranks) / (max over ranks).

Partitions (`--shard`): `alignments` (default; weak scaling: every rank scores its own alignment) or `splits` (the
north-star partition, SURVEY 8e: ONE alignment replicated, its candidate-split set dealt to the ranks by cost class
with batch.shard_indices, one all-gather of the packed scores + status per step; strong scaling).
Other workloads: config5 (batch of 32 simulated 12-taxon alignments x 2035 splits, one device pass per step; the
batch is dealt to the ranks), config3 / config4 (16 / 20 taxa, 1 M bp, all splits, subflattening route).
`--mode dropin` additionally times the literal README loop (README.md:37-41) on the drop-in functions.

Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` for the dominant kernel and
`cpu_baseline` (the oracle's faithful restatement of the reference CPU path on the same table: all 501 splits with
default BLAS threads and again with OMP_NUM_THREADS=1, both as child processes started before any GPU call)."""
import argparse
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
WORKLOADS = {
    #          taxa  sites      alignments  method
    "config2": (10, 100_000, 1, "flattening"),
    "config3": (16, 1_000_000, 1, "subflattening"),
    "config4": (20, 1_000_000, 1, "subflattening"),
    "config5": (12, 100_000, 32, "flattening"),
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


# ----------------------------------------------------------------------------------------------- roofline
def real_work(status, a_arr, n_taxa, patterns):
    """What one launch really computes (from the status words of the last step): sparse half products of the general
    path (smaller side >= 4 taxa; the first one is a scatter of 4 rows and not counted) x non-zeros x 4 block columns."""
    import numpy as np

    if status is None:
        return None
    st = np.asarray(status).reshape(-1, len(a_arr))
    k = np.minimum(a_arr, n_taxa - a_arr)
    general = k >= 4
    halves = int(np.sum(np.maximum((st[:, general] >> 8) - 1, 0)))
    dense_products = int(np.sum(st[:, ~general] >> 8))
    return {"sparse_half_products": halves, "nnz": int(patterns), "block_columns": 4,
            "fma": halves * int(patterns) * 4,
            "small_side_splits": int(np.count_nonzero(~general)) * st.shape[0], "their_dense_G_products": dense_products,
            "note": "general path: W = C^T V / Y = C W on the D non-zeros; smaller sides of <= 3 taxa iterate on the exact "
                    "integer Gram matrix (<= 64 x 64) instead"}


def roofline_block(dom, dom_ms_alone, dom_ms_region, in_flight, phases_per_step, n_taxa, n_sites, a_arr, n_items_per_launch,
                   patterns, lanes, status=None):
    import numpy as np

    k_small = np.minimum(a_arr, n_taxa - a_arr).astype(np.float64)
    per_al = n_items_per_launch / max(len(a_arr), 1)            # alignments per launch
    algo_bytes = float(len(a_arr) * (4.0 * n_sites + 4.0 * 4.0 ** n_taxa)) * per_al       # 4L + 4*4^n per split
    algo_flops = float(np.sum(2.0 * 4.0 ** n_taxa * 4.0 ** k_small)) * per_al            # 2*4^n*4^k per split
    names = {"gram": "k_gram_i8<2,int> (int8-limb MFMA Gram)", "eigen": "k_eig_gv / k_eig_rr (fp64 MFMA)",
             "sparse": "k_sparse_score (one workgroup per split: CSC/CSR lists + 4-wide block in LDS)",
             "scatter": "k_zero_i8 + k_scatter_i8", "reindex": "k_reindex", "subscore": "k_subscore_tri",
             "chain": "k_sparse_slow (persistent workgroups: lists-in-global / all-global / 8-wide forms of the sparse kernel)"}
    sec = dom_ms_alone * 1e-3
    roof = {"kernel": names.get(dom, dom), "launch_ms": dom_ms_alone, "launch_ms_in_timed_region": dom_ms_region,
            "launches_in_flight": in_flight, "phase_ms_per_step": phases_per_step, "traffic": None}
    if dom == "subscore":
        # SURVEY 8(d), subflattening path: per split a Gram 2 m^2 m' on the (3k+1) x (3(n-k)+1) block, Householder
        # tridiagonalisation 4/3 m^3 and Sturm multisection (14 passes x 64 shifts x 2m); bound: fp64 VALU / latency
        m = 3.0 * k_small + 1.0
        mp = 3.0 * (n_taxa - k_small) + 1.0
        flops = float(np.sum(2.0 * m * m * mp + 4.0 / 3.0 * m ** 3 + 14 * 64 * 2.0 * m)) * per_al
        roof.update({"bound": "fp64-valu", "achieved": flops / sec / 1e12, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                     "frac": flops / sec / 1e12 / FP64_MFMA_PEAK_TF, "algorithmic_flops_per_launch": flops,
                     "note": "one wave per split, dependent chains of a few hundred fp64 operations: latency-bound by "
                             "construction (SURVEY 8d: 'fp64 VALU / launch latency; report splits/s and achieved fp64 FLOP/s'); "
                             "peak = the fp64 vector rate (= the fp64 matrix rate on this part)"})
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
    pmc_path = os.path.join(ROOT, "profiles", "r02_pmc_binding.json")
    pmc = None
    if os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path)).get(dom)
        except Exception:
            pmc = None
    if dom in ("gram", "eigen"):
        # Dense route (the north-star pipeline).  Its longest single kernel is the int8-limb MFMA Gram, so `roofline` is
        # that kernel on EXECUTED work: int8 operations one launch issues to the matrix cores (rocprofv3
        # SQ_INSTS_VALU_MFMA_MOPS_I8 x 512, profiles/r02_pmc_dense_route.json - compact matrices, upper triangle, 2 x 2 limb
        # products) / the Gram phase's duration measured live here, against the dense int8 peak; the scatter phase is
        # reported the same way on the HBM bytes its two kernels move (PMC) against the HBM peak.
        dpmc = {}
        dpath = os.path.join(ROOT, "profiles", "r02_pmc_dense_route.json")
        if os.path.exists(dpath):
            try:
                dpmc = json.load(open(dpath))
            except Exception:
                dpmc = {}
        g = dpmc.get("gram") or {}
        gram_s = phases_per_step.get("gram", 0.0) * 1e-3
        ops = g.get("SQ_INSTS_VALU_MFMA_MOPS_I8", 0.0) * 512.0
        roof.update({"kernel": "k_gram_i8_big<2,int> (int8-limb MFMA Gram, 128 x 128 tiles)",
                     "launch_ms": phases_per_step.get("gram"), "bound": "mfma",
                     "achieved": ops / gram_s / 1e12 if ops and gram_s else None, "peak": INT8_MFMA_PEAK_TOPS,
                     "unit": "TOP/s (int8)",
                     "frac": ops / gram_s / 1e12 / INT8_MFMA_PEAK_TOPS if ops and gram_s else None,
                     "executed_int8_ops_per_launch": ops or None,
                     "traffic": g.get("hbm_bytes_per_launch"),
                     "note": "executed work, not SURVEY 8(d)'s fp64 count of the uncompacted problem (kept under survey_8d); "
                             "peak = 256 CUs x 4 SIMDs x one v_mfma_i32_32x32x32_i8 (65536 ops) per 32 cycles x 2.4 GHz"})
        if g.get("SQ_VALU_MFMA_BUSY_CYCLES") and g.get("kernel_cycles_from_GRBM_GUI_ACTIVE_div_8"):
            roof["mfma_busy_frac_pmc"] = g["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * N_CU * g["kernel_cycles_from_GRBM_GUI_ACTIVE_div_8"])
        zs = [dpmc.get("zero") or {}, dpmc.get("scatter") or {}]
        sc_bytes = sum(z.get("hbm_bytes_per_launch", 0.0) for z in zs)
        sc_s = phases_per_step.get("scatter", 0.0) * 1e-3
        if sc_bytes and sc_s:
            roof["scatter_phase"] = {"kernels": "k_zero_i8 + k_scatter_i8", "bound": "hbm", "ms": phases_per_step.get("scatter"),
                                     "hbm_bytes_per_launch_pmc": sc_bytes, "achieved": sc_bytes / sc_s / 1e9,
                                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": sc_bytes / sc_s / 1e9 / HBM_PEAK_GBS}
        roof["eigen_phase"] = {"ms": phases_per_step.get("eigen"),
                               "note": "16-wide block iteration: G V products (k_eig_gv: G streamed from HBM once per "
                                       "product) + per-split Rayleigh-Ritz (k_eig_rr: one wave's 16 x 16 Jacobi, latency-"
                                       "bound); long and short sides as two concurrent pipelines"}
        return roof
    if dom == "sparse" and pmc:
        # What binds k_sparse_score is the LDS array and VALU issue, not HBM.  The binding
        # fraction = LDS-array cycles the launch used (rocprofv3 SQ_LDS_IDX_ACTIVE, summed over the CUs, per launch:
        # profiles/r02_pmc_binding.json, same command) / LDS-array cycles available in the launch's own duration
        # (CUs x duration x clock), with the duration measured live here.
        avail = N_CU * sec * CLOCK_GHZ * 1e9
        lds = pmc["SQ_LDS_IDX_ACTIVE"]
        valu_quad = pmc.get("SQ_ACTIVE_INST_VALU", 0.0)
        roof.update({"bound": "lds", "achieved": lds / sec / 1e12, "peak": N_CU * CLOCK_GHZ * 1e9 / 1e12,
                     "unit": "T LDS-array cycles/s", "frac": lds / avail,
                     "binding": {
                         "lds_array_cycles_per_launch": lds, "lds_array_cycles_available": avail,
                         "lds_busy_frac": lds / avail,
                         "bank_conflict_share": pmc["SQ_LDS_BANK_CONFLICT"] / lds if lds else None,
                         "valu_issue_frac": valu_quad * 4.0 / (4 * avail) if valu_quad else None,
                         "real_work": real_work(status, a_arr, n_taxa, patterns),
                         "cycle_base": f"{N_CU} CUs x launch_ms x {CLOCK_GHZ} GHz (profiles/r02_pmc_binding.json holds the "
                                       "GRBM_GUI_ACTIVE / 8 of the profiled launches next to it)",
                         "source": "profiles/r02_pmc_binding.json"}})
        roof["traffic"] = pmc.get("hbm_bytes_per_launch")
        return roof
    if dom == "chain":
        roof.update({"bound": "lds", "achieved": None, "peak": N_CU * CLOCK_GHZ * 1e9 / 1e12, "unit": "T LDS-array cycles/s",
                     "frac": None,
                     "note": "the slow forms of the sparse kernel (entry lists / every array in global memory): L2-latency "
                             "bound variants of k_sparse_score; no PMC profile of this kernel is committed, so no binding "
                             "fraction is claimed"})
        return roof
    roof.update({"bound": "hbm", "achieved": survey["scatter_phase"]["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": None if dom == "sparse" else survey["scatter_phase"]["frac"],
                 "note": "no PMC profile of this kernel in profiles/r02_pmc_binding.json: binding fraction not computed"})
    return roof


# ----------------------------------------------------------------------------------------------- main
def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--_cpu-leg":
        return _cpu_leg_main(sys.argv[2:])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--shard", default="alignments", choices=["alignments", "splits"])
    ap.add_argument("--mode", default="batched", choices=["batched", "dropin"])
    ap.add_argument("--alignments", type=int, default=0, help="alignments per rank per step (0 = the workload's)")
    ap.add_argument("--lanes", type=int, default=0, help="steps in flight (one sp_ctx + HIP stream + buffers each); 0 = auto")
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
    n_taxa, n_sites, wl_aligns, method_name = WORKLOADS[args.workload]
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
    dev_t = torch.device("cuda", torch.cuda.current_device())

    import ctypes as C

    import splitp_amd as sp
    from splitp_amd import _lib, batch
    from splitp_amd import simulation as sim
    from splitp_amd import synthetic as syn
    from splitp_amd.device import Context

    sp._lib.require_gpu()
    names = syn.taxa_names(n_taxa)
    shard_splits = args.shard == "splits"
    method = sp.Method.flattening if method_name == "flattening" else sp.Method.subflattening
    code = batch._method_code(method, args.route if method_name == "flattening" else "auto")
    use_plan = method_name == "flattening" and args.route in ("auto", "sparse")

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
    if args.alignments > 0:
        n_al_rank, seeds = args.alignments, [1 + rank * args.alignments + a for a in range(args.alignments)]
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

    # ---- lanes: one library context + HIP stream + buffers each ---------------------------------------------------
    # (more lanes with RCCL: the all-gather adds latency to every step, not work)
    n_lanes = args.lanes if args.lanes > 0 else ((5 if dist is not None else 3) if use_plan and args.workload == "config2" else 1)
    if not use_plan:
        n_lanes = 1      # the dense / subflattening routes run in the alignment's own context: one stream, ordered
    plan = batch.SplitPlan(ctx0, taxa_arr, a_arr, n_taxa) if use_plan else None
    lib = ctx0._lib
    taxa_p, a_p = _lib._ptr(taxa_arr, C.c_int32), _lib._ptr(a_arr, C.c_int32)
    al_handles = (C.c_void_p * len(aligns))(*[a.handle.value for a in aligns])

    class Lane:
        def __init__(self):
            self.stream = torch.cuda.Stream(device=dev_t)
            self.ctx = Context(ctx0.device, stream=self.stream.cuda_stream) if use_plan else ctx0
            # per alignment: `per` scores (f64) then `per` status words (int32, padded) - batch.gather_scores' layout
            self.send = torch.zeros(width, dtype=torch.float64, device=dev_t)
            self.recv = torch.zeros(world * width, dtype=torch.float64, device=dev_t) if dist is not None else None
            self.host = torch.zeros(world * width, dtype=torch.float64).pin_memory()
            self.host_np = self.host.numpy()
            self.done = torch.cuda.Event()
            self.busy = False
            base = self.send.data_ptr()
            if n_al_rank == 1 or not use_plan:
                self.sc_p = [C.c_void_p(base + a * width_al * 8) for a in range(n_al_rank)]
                self.st_p = [C.c_void_p(base + a * width_al * 8 + per * 8) for a in range(n_al_rank)]
                self.packed = True
            else:
                # one call scores all alignments: its outputs are alignment-major [n_al][n_mine] scores and status,
                # laid out as two blocks (scores of all alignments, then status of all alignments)
                self.sc_p = [C.c_void_p(base)]
                self.st_p = [C.c_void_p(base + n_al_rank * n_mine * 8)]
                self.packed = False

    lanes = [Lane() for _ in range(n_lanes)]

    def launch(lane):
        with torch.cuda.stream(lane.stream):
            if use_plan:
                if lane.packed:
                    for a in range(n_al_rank):
                        one = (C.c_void_p * 1)(aligns[a].handle.value)
                        _lib.check(lib.sp_score_plan_async(lane.ctx.handle, one, 1, plan.handle, lane.sc_p[a], lane.st_p[a]))
                else:
                    _lib.check(lib.sp_score_plan_async(lane.ctx.handle, al_handles, n_al_rank, plan.handle, lane.sc_p[0],
                                                       lane.st_p[0]))
            elif enum_on_device:
                ctx0.sync_stream_with_torch()
                n_got = C.c_int64()
                for a, al in enumerate(aligns):
                    _lib.check(lib.sp_score_all_splits_shard(al.handle, code, 0, 0, rank if shard_splits else 0,
                                                             world if shard_splits else 1, C.byref(n_got), None,
                                                             lane.sc_p[a], None, lane.st_p[a]))
                    assert n_got.value == n_mine, (n_got.value, n_mine)
            else:
                ctx0.sync_stream_with_torch()
                for a, al in enumerate(aligns):
                    _lib.check(lib.sp_score_splits_async(al.handle, taxa_p, a_p, n_mine, code, lane.sc_p[a], lane.st_p[a]))
            if dist is not None:
                dist.all_gather_into_tensor(lane.recv, lane.send)
                lane.host.copy_(lane.recv, non_blocking=True)
            else:
                lane.host.copy_(lane.send, non_blocking=True)
            lane.done.record()
        lane.busy = True

    def lane_results(lane, r=None):
        """(scores [n_al, n_mine], status [n_al, n_mine]) views of rank r's block in the lane's host buffer."""
        r = rank if r is None else r
        blk = lane.host_np[r * width:(r + 1) * width]
        if lane.packed:
            rows = blk.reshape(n_al_rank, width_al)
            return rows[:, :n_mine], rows[:, per:].view(np.int32)[:, :n_mine]
        sc = blk[:n_al_rank * n_mine].reshape(n_al_rank, n_mine)
        st = blk[n_al_rank * n_mine:].view(np.int32)[:n_al_rank * n_mine].reshape(n_al_rank, n_mine)
        return sc, st

    unresolved = [0]

    def retire(lane):
        """Wait until the lane's scores are on the host (the unit of work is complete) and check the status words of this
        rank's own items: the device chain leaves no bit 1 (not handled) behind; bit 0 (upper estimate) is counted."""
        if not lane.busy:
            return
        lane.done.synchronize()
        lane.busy = False
        _, st = lane_results(lane)
        if (st & 2).any():
            raise SystemExit("bench.py: a split came back unhandled (status bit 1) - the device chain must be complete")
        unresolved[0] += int(np.count_nonzero(st & 1))

    def run(n_steps):
        for i in range(n_steps):
            lane = lanes[i % len(lanes)]
            retire(lane)
            launch(lane)
        for lane in lanes:
            retire(lane)

    # CPU legs finish before the GPU is timed (they would compete for host cores with the launch loop)
    cpu = collect_cpu_legs(legs) if legs else None

    # untimed spin-up: a fresh box needs a few hundred ms of load before clocks, queues and the RCCL channel settle;
    # then the W warmup steps the contract asks for.  (The number of spin-up batches is agreed between the ranks.)
    t_spin = time.perf_counter()
    spin_batch = 50 if args.workload == "config2" else 2
    while args.spinup > 0:
        run(spin_batch)
        go = time.perf_counter() - t_spin < args.spinup
        if dist is not None:
            flag = torch.tensor([1 if go else 0], dtype=torch.int32, device=dev_t)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            go = bool(int(flag.item()))
        if not go:
            break
    run(args.warmup)
    for lane in lanes:
        lane.ctx.enable_timing(True)
        lane.ctx.reset_timing()
    unresolved[0] = 0
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
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
    last = lanes[(args.steps - 1) % len(lanes)]
    scores, status = (x.copy() for x in lane_results(last))

    # the dominant kernel's OWN duration: a few un-overlapped launches on one lane straight after the timed region
    # (inside the region the lanes overlap, so an event-bracketed launch also spans its neighbours' workgroups)
    ph = {k: v for k, v in phases.items() if v[1] > 0}
    dom = max(ph, key=lambda k: ph[k][0])
    alone_n = max(3, min(50, args.steps))
    l0 = lanes[0]
    l0.ctx.enable_timing(True)
    l0.ctx.reset_timing()
    for _ in range(alone_n):
        launch(l0)
        retire(l0)
    alone = l0.ctx.phase_times()
    l0.ctx.enable_timing(False)
    dom_ms_alone = alone[dom][0] / max(alone[dom][1], 1)

    if rank == 0:
        total_items = args.steps * (n_al_rank * n_splits_total if shard_splits else world * items_rank)
        if not shard_splits and wl_aligns > 1 and args.alignments == 0:
            total_items = args.steps * wl_aligns * n_splits_total          # the whole dealt batch
        value = total_items / elapsed
        launches_per_step = ph[dom][1] / args.steps
        roof = roofline_block(dom, dom_ms_alone, ph[dom][0] / ph[dom][1], ph[dom][0] / (elapsed * 1e3),
                              {k: round(v[0] / args.steps, 5) for k, v in ph.items()}, n_taxa, n_sites, a_arr,
                              items_rank / max(launches_per_step, 1e-9), n_patterns, n_lanes, status=status)
        if shard_splits and enum_on_device:
            par = (f"split-sharded x{world}: one alignment replicated, every rank enumerates and scores its share of every size "
                   "class on the device (index mod P), all_gather of scores + status")
        elif shard_splits:
            par = f"split-sharded x{world}: one alignment replicated, candidate splits dealt by cost class, all_gather of scores + status"
        elif world > 1:
            par = f"alignment-sharded x{world}, all_gather of scores + status"
        else:
            par = "single GPU"
        wl_text = {
            "config2": "BASELINE configs[1]: 10-taxon balanced tree (branch 0.05, JC), 100k bp, all 501 splits, "
                       "flattening + fp64 split score (whole sparse route incl. its device-side hand-back chain), scores "
                       "+ status copied to the host every step",
            "config3": "BASELINE configs[2]: 16-taxon balanced tree, 1M bp, all 32751 splits, subflattening route, fp64",
            "config4": "BASELINE configs[3]: 20-taxon balanced tree, 1M bp, all 524267 splits, subflattening route, fp64",
            "config5": "BASELINE configs[4]: batch of 32 device-simulated 12-taxon alignments (100k bp, branch 0.05, JC) x "
                       "all 2035 splits in ONE device pass per step (sparse route + device-side chain); fp64 eigen "
                       "arithmetic - stricter than the config's fp32 wording, which is not built",
        }[args.workload]
        out = {
            "metric": "splits scored/sec (whole node), 10-taxon 100k-bp JC alignment" if args.workload == "config2"
                      else f"splits scored/sec (whole node), {args.workload}",
            "value": value, "unit": "splits/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if (shard_splits or (wl_aligns > 1 and args.alignments == 0)) else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl_text, "workload_key": args.workload, "route": args.route, "alignments_per_rank_per_step": n_al_rank,
                       "lanes": n_lanes, "splits_per_alignment": n_splits_total, "splits_this_rank": n_mine,
                       "patterns": n_patterns, "parallelism": par, "unconverged_splits_in_timed_region": unresolved[0]},
            "roofline": roof,
        }
        if cpu is not None:
            out["cpu_baseline"] = {k: v for k, v in cpu.items() if k != "scores"}
            # end-to-end sanity outside the timed region: the GPU scores of three splits equal the CPU leg's
            for i, ref in cpu["scores"].items():
                j = int(np.nonzero(mine == int(i))[0][0])
                assert abs(ref - scores[0, j]) <= 1e-10, (i, ref, scores[0, j])
        else:
            out["cpu_baseline"] = None
        if args.mode == "dropin" and world == 1 and args.workload == "config2":
            out["dropin"] = dropin_loop(sp, syn, tables[0], names, n_taxa, n_sites, scores[0], cpu)
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
