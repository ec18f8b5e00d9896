# rocprofv3 evidence for the default (sparse) route and the dense route: kernel-trace stats + PMC passes.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r01_final
mkdir -p $OUT
for route in auto dense; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$route -o kt -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --route $route > $OUT/kt_$route.log 2>&1
  cp $(find $OUT/kt_$route -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$route.csv
  tail -1 $OUT/kt_$route.log | cut -c1-200
  find $OUT/kt_$route -name "*kernel_trace.csv" -delete || true
done
for route in auto dense; do
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum"; do
  name=${route}_$(echo $pass | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$name -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --route $route > $OUT/$name.log 2>&1 || { echo "pass $pass failed"; tail -3 $OUT/$name.log; continue; }
  F=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  python3 - "$F" "$route" <<'PY' >> $OUT/pmc_summary.txt
import csv, sys, collections
f, route = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); seen = set()
with open(f) as fh:
    for row in csv.DictReader(fh):
        k = row["Kernel_Name"].split("(")[0][:44]
        if k.startswith("void at::") or "rocclr" in k: continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        key = (row["Dispatch_Id"], k)
        if key not in seen:
            seen.add(key); calls[k] += 1
for k in agg:
    print(route, k, "calls", calls[k], {c: round(v / calls[k], 1) for c, v in agg[k].items()})
PY
  rm -rf $OUT/$name
done
done
cat $OUT/pmc_summary.txt
