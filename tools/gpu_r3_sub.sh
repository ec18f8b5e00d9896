cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "subflat or config3 or config4 or smoke or erickson" > gpurun_out/$1/tests_sub.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/$1/tests_sub.log; grep -n "^E " gpurun_out/$1/tests_sub.log | head
timeout -k 10 100 python tools/gpu_fuzz_sub.py 2>&1 | grep -v amdgpu | tail -3
for wl in config3 config4; do timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/$1/bench_$wl.json 2>/dev/null; python - gpurun_out/$1/bench_$wl.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; print("%s: %.3f ms/step %.4g splits/s launch_ms %.4f frac %.4f" % (d["config"]["workload_key"], d["ms_per_step"], d["value"], r["launch_ms"], r["frac"]))
PY
done
