import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splitp_amd as sp
from splitp_amd import synthetic as syn
from oracle import splitp_oracle as O
from tests.conftest import taxa_names
for n, length, branch, seed in ((8, 30, 0.05, 108), (8, 60, 0.05, 1), (8, 120, 0.05, 1), (8, 400, 0.05, 1), (10, 40, 0.05, 1), (10, 200, 0.05, 1)):
    names = taxa_names(n)
    keys, counts = syn.pattern_table(syn.simulate_sites(n, length, branch, seed=seed))
    splits = list(sp.all_splits(names))
    dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=int(counts.sum()), taxa=names)
    got, st = sp.score_splits(dev, splits, return_status=True)
    ref = sp.score_splits(dev, splits, route="dense")
    bad = [(i, min(len(s[0]), len(s[1])), hex(st[i]), got[i], ref[i]) for i, s in enumerate(splits) if abs(got[i] - ref[i]) > 1e-10]
    byk = {}
    for b in bad: byk[b[1]] = byk.get(b[1], 0) + 1
    print("n", n, "L", length, "D", len(keys), "bad", len(bad), "by k", byk, bad[:2])
