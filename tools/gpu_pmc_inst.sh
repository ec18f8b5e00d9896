#!/bin/bash
# rocprofv3 PMC passes: instruction mix / pipe activity of k_sparse_score (GPU box).  Usage: bash tools/gpu_pmc_inst.sh TAG
set -e
cd $GRAFT_REPO_ROOT
TAG=${1:-r01_inst}
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
grep -oE "SQ_(INSTS|ACTIVE_INST|INST_CYCLES|WAIT|LDS)[A-Z0-9_]*" $OUT/counters_list.txt | sort -u | tr '\n' ' '
echo
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_SALU"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$name -o p -- python3 bench.py --steps 6 --warmup 2 --lanes 1 --spinup 0 --no-cpu-baseline > $OUT/$name.log 2>&1 || { echo "pass $pass failed"; tail -3 $OUT/$name.log; continue; }
  F=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  python3 - "$F" "$pass" <<'PY'
import csv, sys, collections
f, names = sys.argv[1], sys.argv[2].split()
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
seen = set()
with open(f) as fh:
    for row in csv.DictReader(fh):
        k = row["Kernel_Name"].split("(")[0][:40]
        if "sparse" not in k: continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        key = (row["Dispatch_Id"], k)
        if key not in seen:
            seen.add(key); calls[k] += 1
for k in agg:
    print(k, "calls", calls[k], {c: round(v / calls[k], 1) for c, v in agg[k].items()})
PY
done
find $OUT -name "*.csv" -size +5M -delete || true
