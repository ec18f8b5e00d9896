# A/B on one box: the library built from HEAD's sparse.hip (tools/experiments/lib_head.so, built in the container) against
# the working tree's library, bench.py config 2, alternating:  bash tools/gpu_ab_head.sh [steps]
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in head tree; do
    unset SPLITP_OLD_ORDER
    if [ $v = head ]; then export SPLITP_LIB=$GRAFT_REPO_ROOT/tools/experiments/lib_head.so; else unset SPLITP_LIB; fi
    if [ $v = tree_oldorder ]; then export SPLITP_OLD_ORDER=1; fi
    python bench.py --steps ${1:-3000} --warmup 50 --no-cpu-baseline --no-pipeline-block > /tmp/b.json 2>/tmp/b.err || { tail -3 /tmp/b.err; }
    python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print("$v: ms_per_step %.5f launch_ms %.5f" % (d['ms_per_step'], d['roofline']['launch_ms']))
PY
  done
done
