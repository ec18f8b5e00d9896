# Bench a set of variant libraries (built in the container: tools/variant_lib.sh ... tools/experiments/sweep/lib_X.so) against
# the tree's on ONE box: bash tools/gpu_sweep_libs.sh [workload ...]   (default: config5 config2)
cd $GRAFT_REPO_ROOT
wls=${@:-config5 config2}
one() {
  for wl in $wls; do
    steps=1500; [ $wl = config5 ] && steps=30; [ $wl = config4 ] && steps=20; [ $wl = config3 ] && steps=200
    timeout -k 10 200 python bench.py --workload $wl --steps $steps --warmup 10 --no-cpu-baseline --no-pipeline-block > /tmp/b.json 2>/tmp/b.err || { tail -2 /tmp/b.err; continue; }
    python -c "
import json; d=json.load(open('/tmp/b.json')); print('$1 $wl: ms_per_step %.5f launch_ms %.5f unresolved %s' % (d['ms_per_step'], d['roofline']['launch_ms'], d['config']['unconverged_splits_in_timed_region']))"
  done
}
unset SPLITP_LIB; one tree
for lib in tools/experiments/sweep/lib_*.so; do export SPLITP_LIB=$GRAFT_REPO_ROOT/$lib; one $(basename $lib .so); done
unset SPLITP_LIB; one tree
