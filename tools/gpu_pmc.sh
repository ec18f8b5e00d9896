# rocprofv3 PMC passes for the benchmark's kernels (GPU box).  Usage: bash tools/gpu_pmc.sh TAG
set -e
cd $GRAFT_REPO_ROOT
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
grep -c . $OUT/counters_list.txt || true
grep -oE "(SQ_[A-Z0-9_]*MFMA[A-Z0-9_]*|FETCH_SIZE|WRITE_SIZE|SQ_BUSY_CYCLES|SQ_WAVES|GRBM_GUI_ACTIVE|SQ_LDS_BANK_CONFLICT|SQ_LDS_IDX_ACTIVE|TCC_HIT_sum|TCC_MISS_sum|SQ_WAVE_CYCLES|SQ_VALU_MFMA_BUSY_CYCLES|SQ_INSTS_MFMA|SQ_INSTS_VALU_MFMA_MOPS_I8|SQ_INSTS_VALU_MFMA_MOPS_F64)" $OUT/counters_list.txt | sort -u | tr '\n' ' '
echo
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$name -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/$name.log 2>&1 || { echo "pass $pass failed"; tail -5 $OUT/$name.log; continue; }
  F=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  python3 - "$F" "$pass" <<'PY'
import csv, sys, collections
f, names = sys.argv[1], sys.argv[2].split()
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
seen = set()
with open(f) as fh:
    for row in csv.DictReader(fh):
        k = row["Kernel_Name"].split("(")[0][:40]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        key = (row["Dispatch_Id"], k)
        if key not in seen:
            seen.add(key); calls[k] += 1
for k in agg:
    print(k, "calls", calls[k], {c: round(v / calls[k], 1) for c, v in agg[k].items()})
PY
done
find $OUT -name "*.csv" -size +5M -delete || true
