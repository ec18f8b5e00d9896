#!/usr/bin/env python3
"""Benchmark: splits scored/sec on the 10-taxon 100k-bp JC alignment (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One step = one pass of the hot path over one batch: score ALL 501 candidate splits of this rank's alignment
(resident in HBM as a pattern table) with the flattening + fp64 split score - default route: one launch of the in-LDS
sparse kernel - then (N > 1) all-gather every rank's scores over RCCL and copy the scores to pinned host memory.
Steps are issued round-robin on `--lanes` HIP streams (default 3) so the tail of one step's launch, its all-gather
and its D2H copy overlap the next step's kernel; a lane is reused only after its previous step's scores are on the
host and its hand-back flags were checked.  All K steps complete inside the timed region (barrier + device sync on
both sides).  Weak scaling: every rank scores its own alignment(s); value = (splits scored by all ranks) / (max over
ranks of the timed region).

Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` for the dominant
kernel (per-kernel time from HIP events recorded on the launch stream inside the timed region) and
`cpu_baseline` (the oracle's faithful restatement of the reference CPU path, timed on this host on a
bounded sample of the same workload; rank 0, N = 1 only)."""
import argparse
import json
import os
import sys
import time

import numpy as np

# Lanes (below) keep several steps in flight on separate HIP streams next to RCCL's own stream; with the default of 4
# hardware queues two of those streams can share a queue and serialise, so ask the HIP runtime for 8 (read at HIP
# initialisation, hence before torch is imported).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this pool's host driver

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_TAXA = 10
N_SITES = 100_000
BRANCH = 0.05
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_MFMA_PEAK_TF = 78.6     # AMD MI355X FP64 matrix peak; the local guide has no f64 row


def cpu_baseline(table, splits, budget_s=15.0, min_splits=24):
    """Oracle ('port' of the reference's per-pattern Python loops + scipy.linalg.svd) on a
    stratified sample of the same 501 splits, default BLAS threads."""
    from oracle import splitp_oracle as O

    order = []
    stride = 21  # 501 = 3 * 167; stride 21 walks all size classes proportionally
    for off in range(stride):
        order += list(range(off, len(splits), stride))
    t0 = time.perf_counter()
    done = 0
    for i in order:
        m = O.flattening(splits[i], table, "reduced")
        O.split_score(m)
        done += 1
        if done >= min_splits and time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    try:
        from threadpoolctl import threadpool_info

        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    return {"value": done / dt, "unit": "splits/s", "cores": int(threads), "kind": "port",
            "sample": f"{done} of the 501 splits (every 21st, all size classes), FlatFormat.reduced + dense SVD, "
                      f"{dt:.1f} s, host has {os.cpu_count()} logical CPUs"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--alignments", type=int, default=1, help="alignments scored per rank per step")
    ap.add_argument("--lanes", type=int, default=0, help="steps in flight (one HIP stream + buffers each); 0 = 3 on one GPU, 5 with RCCL")
    ap.add_argument("--debug-timeline", action="store_true", help="print host-side retire/launch times of the last steps")
    ap.add_argument("--spinup", type=float, default=1.0, help="seconds of untimed load before the warmup steps")
    ap.add_argument("--no-hipri", action="store_true", help="RCCL stream at normal priority (diagnostic)")
    ap.add_argument("--route", default="auto", choices=["auto", "dense", "sparse"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:   # launched by torch.distributed.run (also with 1 rank)
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        try:
            opts = dist.ProcessGroupNCCL.Options()
            opts.is_high_priority_stream = not args.no_hipri   # the all-gather competes with queued scoring workgroups
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), pg_options=opts)
        except (TypeError, AttributeError):                     # older / newer torch without these options
            dist.init_process_group("nccl")
    else:
        torch.cuda.set_device(0)
    dev_t = torch.device("cuda", torch.cuda.current_device())

    import splitp_amd as sp
    from splitp_amd import _lib, batch
    from splitp_amd import synthetic as syn

    sp._lib.require_gpu()
    names = syn.taxa_names(N_TAXA)
    splits = list(sp.all_splits(names))
    n_splits = len(splits)

    # synthetic input: this rank's alignments, resident in HBM before the timed region
    aligns, tables = [], []
    for a in range(args.alignments):
        seed = 1 + rank * args.alignments + a
        sites = syn.simulate_sites(N_TAXA, N_SITES, BRANCH, seed=seed)
        keys, counts = syn.pattern_table(sites)
        aligns.append(sp.DeviceAlignment.from_arrays(keys, None, N_TAXA, counts=counts, n_sites=N_SITES, taxa=names))
        tables.append((keys, counts))
    taxa_arr, a_arr = batch.encode_splits(splits, aligns[0], N_TAXA)
    code = batch._method_code(sp.Method.flattening, args.route)
    ctx = aligns[0].ctx
    per_rank = args.alignments * n_splits
    # Each lane owns a HIP stream and its own device / pinned-host buffers: per_rank scores (f64) followed by
    # per_rank status words (int32, padded to f64).  Step i runs on lane i % lanes, so the scatter of one step's
    # last workgroups, its all-gather and its D2H copy overlap the next step's kernel (two steps in flight at most
    # per lane pair); a lane is reused only after its previous step's scores are on the host and checked.
    n_stat = (per_rank + 1) // 2
    width = per_rank + n_stat

    class Lane:
        def __init__(self):
            self.stream = torch.cuda.Stream(device=dev_t)
            self.send = torch.zeros(width, dtype=torch.float64, device=dev_t)
            self.recv = torch.zeros(world * width, dtype=torch.float64, device=dev_t) if dist is not None else None
            self.host = torch.zeros(world * width, dtype=torch.float64).pin_memory()
            self.host_np = self.host.numpy()
            self.done = torch.cuda.Event()
            self.busy = False

    # (more lanes with RCCL: the all-gather adds latency to every step, not work.  5, not 4: with 4 lane streams next to
    # RCCL's the step time is 10 % worse than with 3 or 5 - 0.119 vs 0.108 ms under torch.distributed.run on one GPU - two
    # of the streams apparently end up sharing a hardware queue; without RCCL the bad count is 5: 0.122 vs 0.104 ms)
    lanes = [Lane() for _ in range(args.lanes if args.lanes > 0 else (5 if world > 1 else 3))]

    # argument objects of the library calls are built once (the step loop is host-work sensitive: ~50 us of Python per
    # step against ~100 us of GPU work)
    import ctypes as C
    lib = ctx._lib
    taxa_p, a_p = _lib._ptr(taxa_arr, C.c_int32), _lib._ptr(a_arr, C.c_int32)
    al_handles = (C.c_void_p * len(aligns))(*[a.handle.value for a in aligns])
    for lane in lanes:
        lane.stream_p = C.c_void_p(lane.stream.cuda_stream)
        lane.sc_p = [C.c_void_p(lane.send.data_ptr() + a * n_splits * 8) for a in range(len(aligns))]
        lane.st_p = [C.c_void_p(lane.send.data_ptr() + per_rank * 8 + a * n_splits * 4) for a in range(len(aligns))]

    def launch(lane):
        with torch.cuda.stream(lane.stream):
            _lib.check(lib.sp_ctx_set_stream_unordered(ctx.handle, lane.stream_p))
            ctx._stream = lane.stream_p
            if args.route == "auto" and len(aligns) > 1:
                _lib.check(lib.sp_score_splits_multi_async(al_handles, len(aligns), taxa_p, a_p, n_splits, lane.sc_p[0],
                                                           lane.st_p[0]))
            else:
                for a, al in enumerate(aligns):
                    _lib.check(lib.sp_score_splits_async(al.handle, taxa_p, a_p, n_splits, code, lane.sc_p[a], lane.st_p[a]))
            if dist is not None:
                dist.all_gather_into_tensor(lane.recv, lane.send)
                lane.host.copy_(lane.recv, non_blocking=True)
            else:
                lane.host.copy_(lane.send, non_blocking=True)
            lane.done.record()
        lane.busy = True

    def retire(lane):
        """Wait until the lane's scores are on the host (the unit of work is complete), then do the hand-back check
        on this rank's own shard: splits the in-LDS kernel could not take go to the dense route."""
        if not lane.busy:
            return
        lane.done.synchronize()
        lane.busy = False
        mine = lane.host_np[rank * width:(rank + 1) * width]
        st = mine[per_rank:].view(np.int32)[:per_rank]
        if (st & 2).any():
            torch.cuda.synchronize()   # the dense route re-plans into the context's pools: no other lane in flight
            with torch.cuda.stream(lane.stream):
                ctx.sync_stream_with_torch()
                for a, al in enumerate(aligns):
                    batch.finish_async(al, taxa_arr, a_arr, mine[a * n_splits:(a + 1) * n_splits],
                                       st[a * n_splits:(a + 1) * n_splits])
            torch.cuda.synchronize()

    timeline = []

    def run(n_steps):
        for i in range(n_steps):
            lane = lanes[i % len(lanes)]
            if args.debug_timeline:
                ta = time.perf_counter()
                retire(lane)
                tb = time.perf_counter()
                launch(lane)
                timeline.append((ta, tb, time.perf_counter()))
                continue
            retire(lane)
            launch(lane)
        for lane in lanes:
            retire(lane)

    # untimed spin-up: a fresh box needs a few hundred ms of load before clocks, queues and the RCCL channel settle
    # (a 60 ms measurement taken cold is bimodal); then the W warmup steps the contract asks for
    # (the number of spin-up batches is agreed between the ranks: a rank that ran one batch more than its peers would
    # leave 50 all-gathers nobody answers)
    t_spin = time.perf_counter()
    while args.spinup > 0:
        run(50)
        go = time.perf_counter() - t_spin < args.spinup
        if dist is not None:
            flag = torch.tensor([1 if go else 0], dtype=torch.int32, device=dev_t)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            go = bool(int(flag.item()))
        if not go:
            break
    run(args.warmup)
    ctx.enable_timing(True)
    ctx.reset_timing()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if args.debug_timeline and rank == 0:
        base = timeline[-24][0]
        for ta, tb, tc in timeline[-24:]:
            print(f"retire {1e6 * (ta - base):9.1f} -> {1e6 * (tb - base):9.1f}  launch -> {1e6 * (tc - base):9.1f}", file=sys.stderr)
    phases = ctx.phase_times()
    ctx.enable_timing(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev_t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    scores = lanes[(args.steps - 1) % len(lanes)].host_np[:per_rank].copy()

    if rank == 0:
        total_splits = world * per_rank * args.steps
        value = total_splits / elapsed
        # ---- roofline of the dominant kernel (phase) -------------------------------------------
        ph = {k: v for k, v in phases.items() if v[1] > 0}
        dom = max(ph, key=lambda k: ph[k][0])
        dom_ms = ph[dom][0] / ph[dom][1]                      # average duration of one launch (group)
        k_small = np.minimum(a_arr, N_TAXA - a_arr).astype(np.float64)
        algo_flops_gram = float(np.sum(2.0 * 4.0 ** N_TAXA * 4.0 ** k_small))   # 2*4^n*4^k per split
        algo_bytes_scatter = float(n_splits * (4.0 * N_SITES + 4.0 * 4.0 ** N_TAXA))  # 4L + 4*4^n
        kernel_names = {"gram": "k_gram_i8<2,int> (int8-limb MFMA Gram)", "eigen": "k_eig_gv / k_eig_rr (fp64 MFMA)",
                        "sparse": "k_sparse_score (one workgroup per split, CSC/CSR lists + 4-wide block in LDS)",
                        "scatter": "k_zero_i8 + k_scatter_i8", "reindex": "k_reindex"}
        if dom in ("gram", "eigen"):
            achieved = algo_flops_gram / (dom_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": kernel_names[dom],
                    "achieved": achieved, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                    "frac": achieved / FP64_MFMA_PEAK_TF, "traffic": None,
                    "note": "achieved = SURVEY 8(d) algorithmic fp64 Gram flops (2*4^n*4^k per split, 501 splits per "
                            "launch) / measured launch duration; the kernels work on the compacted upper-triangular "
                            "problem (int8 limbs for the Gram), so executed MFMA work is lower (DESIGN.md)"}
        else:
            achieved = algo_bytes_scatter / (dom_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": kernel_names.get(dom, dom), "achieved": achieved, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                    "note": "achieved = SURVEY 8(d) algorithmic bytes of the flattening (4L + 4*4^n per split: every "
                            "site word read once, every dense cell written once) / measured launch duration.  The "
                            "sparse route never materialises the 4^a x 4^b matrix (it keeps the D non-zeros in LDS), so "
                            "its real HBM traffic is ~3 orders of magnitude below this figure and the kernel is bound by "
                            "LDS latency, not HBM (DESIGN.md section 5)"}
        # lanes overlap launches: the event-bracketed duration of one launch spans its neighbours' workgroups too, so
        # the aggregate rate (bytes of all launches / timed region) is reported next to the per-launch one
        in_flight = ph[dom][0] / (elapsed * 1e3)
        roof["launch_ms"] = dom_ms
        roof["launches_in_flight"] = in_flight
        if in_flight > 1.0:
            roof["achieved_per_launch"] = roof["achieved"]
            roof["achieved"] = roof["achieved"] * in_flight
            roof["frac"] = roof["achieved"] / roof["peak"]
            roof["note"] += ("  With %d lanes %.2f launches are in flight on average: `achieved` is the aggregate over "
                             "concurrent launches (= bytes per launch x launches / timed region), `achieved_per_launch` "
                             "uses the event-bracketed launch duration `launch_ms`." % (len(lanes), in_flight))
        roof["phase_ms_per_step"] = {k: round(v[0] / args.steps, 5) for k, v in ph.items()}
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                entry = json.load(open(pmc)).get(dom)
                roof["traffic"] = entry["bytes_per_launch"] if entry else None   # HBM bytes per launch (rocprofv3 PMC, profiles/)
            except Exception:
                pass
        out = {
            "metric": "splits scored/sec (whole node), 10-taxon 100k-bp JC alignment",
            "value": value, "unit": "splits/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 10-taxon balanced tree (branch 0.05, JC), 100k bp, all 501 "
                                   "splits, flattening + fp64 split score, scores copied to host every step",
                       "route": args.route,
                       "alignments_per_rank_per_step": args.alignments, "lanes": len(lanes), "splits_per_alignment": n_splits,
                       "patterns": int(len(tables[0][0])), "parallelism": f"alignment-sharded x{world}, all_gather of scores"
                       if world > 1 else "single GPU"},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            table = syn.table_as_dict(tables[0][0], tables[0][1], N_TAXA, total=N_SITES)
            out["cpu_baseline"] = cpu_baseline(table, splits, budget_s=args.cpu_budget)
            # cheap end-to-end sanity: the GPU scores of the sampled splits agree with the oracle
            from oracle import splitp_oracle as O
            for i in (0, 250, 500):
                ref = O.split_score(O.flattening(splits[i], table, "reduced"))
                assert abs(ref - scores[i]) <= 1e-10, (i, ref, scores[i])
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
