cd $GRAFT_REPO_ROOT
run() { name=$1; shift; "$@" > /tmp/b.json 2>/tmp/b.err || { echo "$name FAILED"; tail -5 /tmp/b.err; return; }; python - "$name" <<'PY'
import json, sys
d=json.load(open('/tmp/b.json'))
print("%s: value %.4g ms_per_step %.4f launch_ms %.4f lanes %s host_us %.1f" % (sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['config'].get('lanes'), d.get('host_us_per_step', -1)))
PY
}
for rep in 1 2; do
run config3 python bench.py --workload config3 --steps 100 --warmup 5 --no-cpu-baseline
run config3_l1 python bench.py --workload config3 --steps 100 --warmup 5 --no-cpu-baseline --lanes 1
run config3_l3 python bench.py --workload config3 --steps 100 --warmup 5 --no-cpu-baseline --lanes 3
run config4 python bench.py --workload config4 --steps 10 --warmup 2 --no-cpu-baseline
run config4_l1 python bench.py --workload config4 --steps 10 --warmup 2 --no-cpu-baseline --lanes 1
done
run dense python bench.py --steps 30 --warmup 3 --no-cpu-baseline --route dense
run dist1_config4_splits python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --workload config4 --shard splits --steps 10 --warmup 2 --no-cpu-baseline
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "bench" 2>&1 | tail -2
