# Config 3 / 4 with the two-splits-a-wave kernel and the one-split-a-wave kernel (GPU box):  bash tools/gpu_subpair_bench.sh
cd $GRAFT_REPO_ROOT
for rep in ${REPS:-1 2}; do
for pair in 1 0; do
  export SPLITP_SUBSCORE_PAIR=$pair
  for wl in config3 config4; do
    steps=10; [ $wl = config3 ] && steps=100
    timeout -k 10 300 python bench.py --workload $wl --steps $steps --warmup 3 --no-cpu-baseline --no-pipeline-block > /tmp/b.json 2>/tmp/b.err || { tail -3 /tmp/b.err; exit 1; }
    python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print("pair $pair $wl: ms_per_step %.4f launch_ms %.4f" % (d['ms_per_step'], d['roofline']['launch_ms']))
PY
  done
done
done
