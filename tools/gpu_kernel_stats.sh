# Per-kernel time of any command under rocprofv3 (GPU box): bash tools/gpu_kernel_stats.sh OUTDIR python3 tools/whatever.py ...
# (the program itself after the output directory - never a shell or env wrapper: the profiler initialises the GPU first)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- "$@" > $OUT/trace.log 2>&1 || tail -5 $OUT/trace.log
grep -v amdgpu.ids $OUT/trace.log | tail -5
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 - $OUT/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.3f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs']) / 1e3:10.2f} us  total {float(r['TotalDurationNs']) / 1e6:9.3f} ms  {float(r['TotalDurationNs']) / tot * 100:5.1f} %")
PY
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +3M -delete
