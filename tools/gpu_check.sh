# check of a kernel change (GPU box): bash tools/gpu_check.sh TAG [quick] - GPU tests, config-5 size classes, config 5 and config 2 bench
cd $GRAFT_REPO_ROOT
TAG=${1:-r3x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ "$2" != "quick" ]; then
  timeout -k 10 500 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $OUT/tests.log
  [ $rc -ne 0 ] && { grep -n "^E " $OUT/tests.log | head -20; exit 1; }
fi
timeout -k 10 200 python tools/gpu_cfg5_classes.py > $OUT/cfg5_classes.txt 2>&1; echo "classes rc=$?"; grep -v amdgpu.ids $OUT/cfg5_classes.txt
timeout -k 10 200 python bench.py --workload config5 --steps 10 --warmup 2 --no-cpu-baseline --no-pipeline-block > $OUT/bench_config5.json 2> $OUT/bench_config5.err; echo "bench5 rc=$?"
python - $OUT/bench_config5.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("config5: %.3f ms/step  %.3g splits/s  unconverged %s" % (d["ms_per_step"], d["value"], d["config"]["unconverged_splits_in_timed_region"]))
PY
timeout -k 10 200 python bench.py --steps 2000 --warmup 50 --no-cpu-baseline --no-pipeline-block > $OUT/bench_config2.json 2> $OUT/bench_config2.err; echo "bench2 rc=$?"
python - $OUT/bench_config2.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("config2: %.5f ms/step  %.4g splits/s  launch_ms %.5f" % (d["ms_per_step"], d["value"], d["roofline"]["launch_ms"]))
PY
