"""Run-to-run determinism of the sparse route on small random tables (GPU box): every table is scored three times through
fresh device alignments; any bit difference is printed.  python tools/gpu_determinism.py SEED TRIALS NMAX"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splitp_amd as sp
from tests.conftest import taxa_names
from tests.test_gpu_parity import _copy_mutate_table
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ntr = int(sys.argv[2]) if len(sys.argv) > 2 else 300
nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rng = np.random.default_rng(seed0)
bad = 0; t0 = time.time()
for trial in range(ntr):
    n = int(rng.integers(4, nmax + 1)); length = int(rng.choice([10, 60, 400, 2500, 20000])); letters = int(rng.choice([2, 3, 4, 4]))
    keys, counts = _copy_mutate_table(rng, n, length, letters)
    if trial % 3 == 0: counts = counts * int(rng.choice([300, 70_000]))   # counts beyond 16 bits: pieces of one cell in the lists
    names = taxa_names(n)
    splits = list(sp.all_splits(names))
    if len(splits) > 200:
        idx = rng.choice(len(splits), size=200, replace=False); splits = [splits[i] for i in sorted(idx)]
    res = []
    for rep in range(3):
        dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=int(counts.sum()), taxa=names)
        s, st = sp.score_splits(dev, splits, return_status=True)
        res.append((s.copy(), st.copy()))
    for rep in (1, 2):
        d = np.nonzero((res[rep][0] != res[0][0]) & ~(np.isnan(res[rep][0]) & np.isnan(res[0][0])))[0]
        if len(d):
            bad += 1
            i = int(d[0])
            print("DIFF trial", trial, "n", n, "L", length, "letters", letters, "D", len(keys), "rep", rep, "splits differing", len(d), "first", i,
                  [len(x) for x in splits[i]], "%.17g vs %.17g" % (res[0][0][i], res[rep][0][i]), hex(int(res[0][1][i])), hex(int(res[rep][1][i])))
            break
print("seed", seed0, "trials", ntr, "tables with run-to-run differences", bad, "%.0f s" % (time.time() - t0))
