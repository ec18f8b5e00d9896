"""Timing of the 12-taxon shapes (BASELINE config 5): LDS form (20 k sites) and global-memory form (100 k sites)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splitp_amd as sp
from splitp_amd import simulation as sim, synthetic as syn
n = 12
names = syn.taxa_names(n)
splits = list(sp.all_splits(names))
for L in (20_000, 100_000):
    dev = sim.generate_device_alignment(syn.balanced_tree(n), sim.JukesCantor(), L, seed=5, branch_length=0.05)
    dev.taxa = tuple(names)
    from splitp_amd import batch, _lib
    taxa_arr, a_arr = batch.encode_all_splits(n)          # (the Python split objects cost more than the scoring)
    batch.score_encoded(dev, taxa_arr, a_arr, _lib.SP_METHOD_FLATTENING)
    dev.ctx.enable_timing(True)
    for rep in range(2):
        dev.ctx.reset_timing()
        t0 = time.perf_counter()
        s, st = batch.score_encoded(dev, taxa_arr, a_arr, _lib.SP_METHOD_FLATTENING)
        dt = time.perf_counter() - t0
        ph = {k: (round(v[0], 3), v[1]) for k, v in dev.ctx.phase_times().items() if v[1]}
        print("L", L, "D", dev.info()["D"], "wall %.2f ms" % (dt * 1e3), ph, "splits/s %.0f" % (len(splits) / dt))
