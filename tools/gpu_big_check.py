"""Big-table form of the sparse route: forced on tables the list kernels take (agreement), then on a table only it can take."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splitp_amd as sp
from splitp_amd import synthetic as syn
from oracle import splitp_oracle as O
def run(n, L, branch, seed, nsplits=None):
    names = syn.taxa_names(n)
    sites = syn.simulate_sites(n, L, branch, seed=seed)
    keys, counts = syn.pattern_table(sites)
    dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=L, taxa=names)
    splits = list(sp.all_splits(names))
    if nsplits: splits = splits[:: max(1, len(splits) // nsplits)]
    return names, keys, counts, dev, splits
for n, L, br in ((10, 100_000, 0.05), (12, 100_000, 0.05)):
    names, keys, counts, dev, splits = run(n, L, br, 1, 400)
    base = sp.score_splits(dev, splits)
    os.environ["SPLITP_FORCE_BIG"] = "1"
    t0 = time.perf_counter(); big, st = sp.score_splits(dev, splits, return_status=True); dt = time.perf_counter() - t0
    again = sp.score_splits(dev, splits)
    del os.environ["SPLITP_FORCE_BIG"]
    print(f"n={n} D={len(keys)} splits={len(splits)}: max |big - lists| = {np.abs(big - base).max():.2e}, repeatable {np.array_equal(big, again)}, "
          f"flags {sorted(set((st & 3).tolist()))}, its {sorted(set((st >> 8).tolist()))[:6]}, {dt*1e3:.1f} ms")
# a table only the big form can take: 12 taxa, 1 M sites, longer branches
names, keys, counts, dev, splits = run(12, 1_000_000, 0.08, 2, 120)
t0 = time.perf_counter(); got, st = sp.score_splits(dev, splits, return_status=True); dt = time.perf_counter() - t0
print(f"n=12 L=1M D={len(keys)} splits={len(splits)}: {dt*1e3:.1f} ms, flags {sorted(set((st & 3).tolist()))}, its {sorted(set((st >> 8).tolist()))[:8]}")
worst = 0.0; nchk = 0
for i, spl in enumerate(splits):
    if min(len(spl[0]), len(spl[1])) > 3: continue
    M = O.reduced_flattening_packed(keys, counts.astype(np.float64), 12, [names.index(t) for t in spl[0]], [names.index(t) for t in spl[1]])[0]
    worst = max(worst, abs(O.dense_split_score(M) - got[i])); nchk += 1
print("oracle checks", nchk, "worst %.2e" % worst)
