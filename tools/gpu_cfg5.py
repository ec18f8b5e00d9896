"""BASELINE config 5 shape: simulated 12-taxon alignments x all 2035 splits - what do the routes do?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splitp_amd as sp
from splitp_amd import simulation as sim, synthetic as syn
from oracle import splitp_oracle as O
n = 12
names = syn.taxa_names(n)
splits = list(sp.all_splits(names))
print("splits", len(splits))
for L in (20_000, 100_000):
    dev = sim.generate_device_alignment(syn.balanced_tree(n), sim.JukesCantor(), L, seed=5, branch_length=0.05)
    dev.taxa = tuple(names)
    print("L", L, dev.info())
    for route in ("auto", "dense"):
        try:
            t0 = time.perf_counter()
            s, st = sp.score_splits(dev, splits, return_status=True, route=route)
            dt = time.perf_counter() - t0
            print(" route", route, "ok %.1f ms" % (dt * 1e3), "its", np.unique(st >> 8)[:8], "flags", np.unique(st & 3), "min/max", s.min(), s.max())
        except Exception as e:
            print(" route", route, "FAILED:", str(e)[:300])
    keys, w, cnt = dev.fetch()
    for i in (0, 700, 2034):
        oa = [names.index(t) for t in splits[i][0]]; ob = [names.index(t) for t in splits[i][1]]
        M = O.reduced_flattening_packed(keys, cnt.astype(np.float64), n, oa, ob)[0]
        try:
            print("  split", i, len(oa), M.shape, "oracle", O.dense_split_score(M), "gpu", s[i])
        except Exception as e:
            print("  oracle fail", e)
