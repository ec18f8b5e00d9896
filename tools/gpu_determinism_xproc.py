"""Cross-process determinism: the tables of tools/gpu_fuzz_long.py (same generator) scored on the default route; all scores
are written to OUT.npy - run it twice (two processes) and compare the files.  python tools/gpu_determinism_xproc.py SEED TRIALS NMAX OUT"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splitp_amd as sp
from tests.conftest import taxa_names
from tests.test_gpu_parity import _copy_mutate_table
seed0, ntr, nmax, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
rng = np.random.default_rng(seed0)
allv, meta = [], []
for trial in range(ntr):
    n = int(rng.integers(4, nmax + 1)); length = int(rng.choice([10, 60, 400, 2500, 20000])); letters = int(rng.choice([2, 3, 4, 4]))
    keys, counts = _copy_mutate_table(rng, n, length, letters)
    if trial % 7 == 0: counts = counts * int(rng.choice([300, 70_000]))
    names = taxa_names(n)
    if n <= 7:
        splits = list(sp.all_splits(names))
    else:
        splits = []
        for _ in range(16):
            k = int(rng.integers(2, n - 1)); left = sorted(rng.choice(n, size=k, replace=False).tolist())
            splits.append((tuple(names[t] for t in left), tuple(names[t] for t in range(n) if t not in left)))
    dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=int(counts.sum()), taxa=names)
    got, st = sp.score_splits(dev, splits, return_status=True)
    for i in range(len(splits)):
        allv.append(got[i]); meta.append((trial, n, len(keys), i, int(st[i])))
np.save(out, np.array(allv)); np.save(out + ".meta.npy", np.array(meta))
