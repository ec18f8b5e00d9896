#!/bin/bash
# rocprofv3 evidence for the default (sparse) route: kernel-trace stats of the bench command + PMC passes.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r01f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
cut -c1-300 $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --steps 300 --warmup 30 --spinup 0.3 --no-cpu-baseline > $OUT/kt.log 2>&1
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_sparse_route.csv
tail -1 $OUT/kt.log | cut -c1-200
find $OUT/kt -name "*kernel_trace.csv" -delete || true
# un-overlapped kernel duration (one lane) for comparison
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt1 -o kt -- python3 bench.py --steps 300 --warmup 30 --spinup 0.3 --lanes 1 --no-cpu-baseline > $OUT/kt1.log 2>&1
cp $(find $OUT/kt1 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_sparse_route_1lane.csv
find $OUT/kt1 -name "*kernel_trace.csv" -delete || true
: > $OUT/pmc_summary.txt
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$name -o p -- python3 bench.py --steps 6 --warmup 2 --spinup 0 --lanes 1 --no-cpu-baseline > $OUT/$name.log 2>&1 || { echo "pass $pass failed"; tail -3 $OUT/$name.log; continue; }
  F=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  python3 - "$F" <<'PY' >> $OUT/pmc_summary.txt
import csv, sys, collections
f = sys.argv[1]
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
    print("sparse-route", k, "calls", calls[k], {c: round(v / calls[k], 1) for c, v in agg[k].items()})
PY
  rm -rf $OUT/$name
done
cat $OUT/pmc_summary.txt
rm -rf $OUT/kt $OUT/kt1
