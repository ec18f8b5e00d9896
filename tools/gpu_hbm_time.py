"""Time the global-memory form of the sparse kernel on the 10-taxon benchmark table (debug: LDS cap forced low)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splitp_amd as sp
from splitp_amd import synthetic as syn
n, L = 10, 100_000
names = syn.taxa_names(n)
keys, counts = syn.pattern_table(syn.simulate_sites(n, L, 0.05, seed=1))
dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=L, taxa=names)
splits = list(sp.all_splits(names))
ref = sp.score_splits(dev, splits)
dev.ctx.enable_timing(True)
for rep in range(3):
    dev.ctx.reset_timing()
    t0 = time.perf_counter()
    s, st = sp.score_splits(dev, splits, return_status=True)
    dt = time.perf_counter() - t0
    ph = {k: v for k, v in dev.ctx.phase_times().items() if v[1]}
    print("wall %.2f ms" % (dt * 1e3), ph, "max diff", np.abs(s - ref).max(), "flags", np.unique(st & 3))
