# A/B of two builds on one box, alternating: the library at $1 (built in the container, e.g. tools/experiments/lib_x.so)
# against the tree's, bench.py workloads $2... (default config2 config5; "dense" = config 2 on the dense route):
#   bash tools/gpu_ab_lib.sh tools/experiments/lib_x.so [workload ...]
cd $GRAFT_REPO_ROOT
other=$GRAFT_REPO_ROOT/$1; shift
wls=${@:-config2 config5}
for rep in 1 2 3; do
  for v in other tree; do
    if [ $v = other ]; then export SPLITP_LIB=$other; else unset SPLITP_LIB; fi
    for wl in $wls; do
      steps=3000; extra=""
      [ $wl = config5 ] && steps=40; [ $wl = config3 ] && steps=200; [ $wl = config4 ] && steps=20
      name=$wl
      if [ $wl = dense ]; then name=config2; steps=60; extra="--route dense"; fi
      timeout -k 10 300 python bench.py --workload $name $extra --steps $steps --warmup 20 --no-cpu-baseline --no-pipeline-block > /tmp/b.json 2>/tmp/b.err || { tail -3 /tmp/b.err; }
      python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print("$v $wl: ms_per_step %.5f launch_ms %.5f phases %s" % (d['ms_per_step'], d['roofline']['launch_ms'], d['roofline'].get('phase_ms_per_step')))
PY
    done
  done
done
