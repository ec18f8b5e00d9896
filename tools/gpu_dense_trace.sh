# dense route diagnostics (GPU box):  bash tools/gpu_dense_trace.sh TAG
# rocprofv3 kernel trace of a few dense-route steps -> the launches of ONE step in order with their durations, kernel
# stats, and the histogram of products (status >> 8) per size class.
set -e
cd $GRAFT_REPO_ROOT
TAG=${1:-dense}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 bench.py --steps 6 --warmup 2 --lanes 1 --no-cpu-baseline --spinup 0.2 --route dense > $OUT/trace.log 2>&1 || tail -5 $OUT/trace.log
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 - $OUT <<'PY'
import csv, sys, glob
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last full step: walk back from the end to the last k_reindex launch
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("void k_reindex") or n.startswith("k_reindex")]
start = idx[-1]
t0 = int(rows[start]["Start_Timestamp"])
with open(out + "/one_step.txt", "w") as fh:
    for r in rows[start:start + 40]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        line = "%8.1f us  +%7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:60])
        print(line); fh.write(line + "\n")
PY
python3 - <<'PY' | tee $OUT/iters.txt
import sys, collections
import numpy as np
sys.path.insert(0, '.')
import splitp_amd as sp
from splitp_amd import synthetic as syn, batch, _lib
n, L = 10, 100_000
names = syn.taxa_names(n)
keys, counts = syn.pattern_table(syn.simulate_sites(n, L, 0.05, seed=1))
dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=L, taxa=names)
taxa_arr, a_arr = sp.encode_all_splits(n)
sc, st = batch.score_encoded(dev, taxa_arr, a_arr, _lib.SP_METHOD_FLATTENING_DENSE)
k = np.minimum(a_arr, n - a_arr)
for kk in (2, 3, 4, 5):
    it = st[k == kk] >> 8
    print("dense route k", kk, "splits", int((k == kk).sum()), "products:", dict(sorted(collections.Counter(it.tolist()).items())))
PY
find $OUT -name "*.db" -delete || true
find $OUT -name "*.csv" -size +3M -delete || true
