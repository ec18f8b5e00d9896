# full round-3 check (GPU box): tests, default bench line (short, with the CPU legs skipped), config 5, group / lane variants
cd $GRAFT_REPO_ROOT
TAG=${1:-r3x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $OUT/tests.log
[ $rc -ne 0 ] && { grep -n "^E " $OUT/tests.log | head -30; exit 1; }
for g in 1 4 8; do
  timeout -k 10 200 python bench.py --steps 2000 --warmup 50 --no-cpu-baseline --group $g > $OUT/bench_config2_g$g.json 2> $OUT/bench_config2_g$g.err; echo "bench2 g=$g rc=$?"
  python - $OUT/bench_config2_g$g.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; ns=d.get("north_star_pipeline") or {}
print("config2: %.5f ms/step  %.4g splits/s  launch_ms %.5f host_us/step %.2f frac %s | dense pipeline %.3f ms %s" % (d["ms_per_step"], d["value"], r["launch_ms"], d["host_us_per_step"], r["frac"], ns.get("ms_per_step", 0), ns.get("phase_ms_per_step")))
PY
done
timeout -k 10 200 python bench.py --workload config5 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_config5.json 2> $OUT/bench_config5.err; echo "bench5 rc=$?"
python - $OUT/bench_config5.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("config5: %.3f ms/step  %.3g splits/s  unconverged %s phases %s" % (d["ms_per_step"], d["value"], d["config"]["unconverged_splits_in_timed_region"], d["roofline"]["phase_ms_per_step"]))
PY
