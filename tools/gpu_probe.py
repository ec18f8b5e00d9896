"""Scratch probe (GPU box): iteration counts and per-phase times per split size class."""
import sys, os, time
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
ctx = dev.ctx
import sys as _s
CODE = int(_s.argv[1]) if len(_s.argv) > 1 else 0
for k in (2, 3, 4, 5, 0):
    sub = [s for s in splits if min(len(s[0]), len(s[1])) == k] if k else splits
    taxa_arr, a_arr = batch.encode_splits(sub, dev, n)
    sc, st = batch.score_encoded(dev, taxa_arr, a_arr, CODE)
    its = st >> 8
    ctx.enable_timing(True); ctx.reset_timing()
    t0 = time.perf_counter()
    for _ in range(20):
        batch.score_encoded(dev, taxa_arr, a_arr, CODE)
    dt = (time.perf_counter() - t0) / 20
    ph = ctx.phase_times(); ctx.enable_timing(False)
    ref_sc, _ = batch.score_encoded(dev, taxa_arr, a_arr, 2)
    print(f"maxdiff vs dense {np.abs(sc-ref_sc).max():.2e}", end=" ")
    print(f"k={k} splits={len(sub)} iters min/mean/max = {its.min()}/{its.mean():.2f}/{its.max()} flagged={int((st&1).sum())} "
          f"wall {dt*1e3:.3f} ms  " + " ".join(f"{p}={v[0]/max(v[1],1):.3f}" for p, v in ph.items() if v[1]))
