# every bench.py workload / partition once, short (GPU box): bash tools/gpu_bench_variants.sh OUTDIR
cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/variants}
mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*"; timeout -k 10 300 "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED rc=$?"; tail -5 $OUT/$name.err; }; python3 - $OUT/$name.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print("   value %.4g %s  ms_per_step %.4f  n_gpus %d  scaling %s  bound %s frac %s  kernel %s launch_ms %.4f" % (
        d["value"], d["unit"], d["ms_per_step"], d["n_gpus"], d["scaling"], r.get("bound"), r.get("frac"), r["kernel"][:40], r["launch_ms"]))
    print("   parallelism:", d["config"]["parallelism"], "| phases", r["phase_ms_per_step"])
except Exception as e:
    print("   no JSON:", e)
PY
}
run config2 python bench.py --steps 2000 --warmup 100 --no-cpu-baseline
run config2_dense python bench.py --steps 30 --warmup 3 --no-cpu-baseline --route dense
run config2_4al python bench.py --steps 300 --warmup 10 --no-cpu-baseline --alignments 4
run config5 python bench.py --workload config5 --steps 20 --warmup 2 --no-cpu-baseline
run config3 python bench.py --workload config3 --steps 50 --warmup 3 --no-cpu-baseline
run config4 python bench.py --workload config4 --steps 10 --warmup 2 --no-cpu-baseline
run dist1_alignments python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 500 --warmup 20 --no-cpu-baseline
run dist1_splits python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 500 --warmup 20 --no-cpu-baseline --shard splits
run dist1_config4_splits python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --workload config4 --shard splits --steps 10 --warmup 2 --no-cpu-baseline
echo "== --gpus 2 on this box (must fail loudly)"
timeout 120 python bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/gpus2.json 2> $OUT/gpus2.err; echo "rc=$?"; grep -h "rank(s)" $OUT/gpus2.err | head -2
