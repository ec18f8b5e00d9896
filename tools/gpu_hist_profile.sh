# Histogram pass under rocprofv3 (GPU box): bash tools/gpu_hist_profile.sh OUTDIR N_TAXA N_SITES [form] [three]
# kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes; prints per-kernel average duration and HBM bytes.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$1; N=$2; L=$3; FORM=${4:-auto}; THREE=${5:-}
mkdir -p $OUT
CMD="python3 tools/hist_driver.py $N $L 20 $FORM $THREE"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1 || tail -5 $OUT/trace.log
grep "^hist" $OUT/trace.log
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o p -- $CMD > $OUT/$c.log 2>&1 || tail -5 $OUT/$c.log
  cp $(find $OUT/$c -name "*counter_collection.csv" | head -1) $OUT/$c.csv
done
python3 - $OUT <<'PY'
import csv, sys, collections
out = sys.argv[1]
dur = {}
for row in csv.DictReader(open(out + "/kernel_stats.csv")):
    dur[row["Name"]] = (int(row["Calls"]), float(row["AverageNs"]) / 1e3)
byt = collections.defaultdict(lambda: [0.0, 0.0, 0])
for ci, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    for row in csv.DictReader(open(out + "/" + c + ".csv")):
        if row["Counter_Name"] != c: continue
        k = row["Kernel_Name"]
        byt[k][ci] += float(row["Counter_Value"])
        if ci == 0: byt[k][2] += 1
print("kernel | calls | avg us | HBM MB read per launch (FETCH_SIZE KB x 2, MI355X_MICROARCH.md gfx950 correction) | MB written (WRITE_SIZE KB) | GB/s")
tot_us = tot_b = 0.0
for k, (calls, us) in sorted(dur.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    f, w, n = byt.get(k, [0, 0, 0])
    n = max(n, 1)
    rd, wr = 2.0 * f / n * 1024 / 1e6, w / n * 1024 / 1e6
    print(f"{k[:56]:56s} {calls:6d} {us:9.2f} us  read {rd:9.3f} MB  write {wr:9.3f} MB  {(rd + wr) * 1e6 / (us * 1e-6) / 1e9 if us else 0:8.1f} GB/s")
    if "hist" in k or "rs_" in k or "os_" in k or "rle" in k or "bins" in k or "pack" in k or "scan" in k or "widen" in k or "weights" in k:
        per_al = calls / 21.0     # 20 timed alignments + 1 first call
        tot_us += us * per_al
        tot_b += (rd + wr) * per_al
print(f"histogram pass, all its kernels: {tot_us:.1f} us of kernel time and {tot_b:.2f} MB of HBM traffic per alignment = {tot_b * 1e6 / (tot_us * 1e-6) / 1e9 if tot_us else 0:.1f} GB/s while a kernel runs")
PY
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +3M -delete
