"""Timing of the mutual-information route (config 2 table, all 501 splits) next to the flattening score."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splitp_amd as sp
from splitp_amd import synthetic as syn, batch, _lib
n, L = 10, 100_000
names = syn.taxa_names(n)
sites = syn.simulate_sites(n, L, 0.05, seed=1)
keys, counts = syn.pattern_table(sites)
dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=L, taxa=names)
splits = list(sp.all_splits(names))
taxa_arr, a_arr = batch.encode_splits(splits, dev, n)
for name, code in (("mutual_information", _lib.SP_METHOD_MUTUAL_INFORMATION), ("flattening", _lib.SP_METHOD_FLATTENING), ("subflattening", _lib.SP_METHOD_SUBFLATTENING)):
    batch.score_encoded(dev, taxa_arr, a_arr, code)
    dev.ctx.enable_timing(True); dev.ctx.reset_timing()
    t0 = time.perf_counter(); reps = 20
    for _ in range(reps):
        sc, st = batch.score_encoded(dev, taxa_arr, a_arr, code)
    dt = (time.perf_counter() - t0) / reps
    ph = {k: round(v[0] / max(v[1], 1), 4) for k, v in dev.ctx.phase_times().items() if v[1]}
    dev.ctx.enable_timing(False)
    print(f"{name}: {dt*1e3:.3f} ms per call of {len(splits)} splits ({len(splits)/dt:.3e} splits/s, synchronous API)", ph)
