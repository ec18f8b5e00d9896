# The driver's 20-step command with the phase timers' event pools created late (old order: between warmup and the timed
# region) against early (ahead of the spin-up), alternating on one box; retire times of the timed region printed:
#   bash tools/gpu_prearm_ab.sh OUTDIR
cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/prearm}
mkdir -p $OUT
one() {  # name, env LATE, extra args...
  name=$1; late=$2; shift 2
  SPLITP_BENCH_LATE_EVENTS=$late SPLITP_BENCH_TRACE=1 timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline-block "$@" > /tmp/b.json 2>/tmp/b.err || { tail -3 /tmp/b.err; return; }
  python - <<PY | tee -a $OUT/ab.txt
import json
d=json.load(open('/tmp/b.json'))
r=d.get('retire_ms') or []
print("$name $*: value %.4g ms_per_step %.5f host_us %.1f retire_ms %s" % (d['value'], d['ms_per_step'], d.get('host_us_per_step', -1), r))
PY
}
for rep in 1 2 3; do
  one late 1
  one early 0
done
for ln in 2 4; do
  one early 0 --lanes $ln
  one early 0 --lanes $ln
done
