"""Timing of the big-table form: 12 taxa, 1 M sites at branch length 0.08 (124 k patterns), all 2035 splits."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splitp_amd as sp
from splitp_amd import synthetic as syn, batch, _lib
n, L = 12, 1_000_000
names = syn.taxa_names(n)
sites = syn.simulate_sites(n, L, 0.08, seed=2)
keys, counts = syn.pattern_table(sites)
dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=L, taxa=names)
taxa_arr, a_arr = batch.encode_all_splits(n)
for rep in range(3):
    t0 = time.perf_counter()
    sc, st = batch.score_encoded(dev, taxa_arr, a_arr, _lib.SP_METHOD_FLATTENING)
    dt = time.perf_counter() - t0
    print(f"D={len(keys)} splits={len(a_arr)}: {dt*1e3:.1f} ms ({len(a_arr)/dt:.0f} splits/s), flags {sorted(set((st & 3).tolist()))}, its max {int((st >> 8).max())}", flush=True)
