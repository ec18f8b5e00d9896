"""Randomised hunt on tables that do not fit the LDS form (lists-in-global and all-global forms of the sparse kernel):
auto route against the dense route (independent kernels) and, where the smaller side allows, against the oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splitp_amd as sp
from splitp_amd import synthetic as syn
from oracle import splitp_oracle as O
from tests.conftest import taxa_names
from tests.test_gpu_parity import _copy_mutate_table
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ntr = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rng = np.random.default_rng(seed0)
bad = 0; checked = 0; t0 = time.time()
for trial in range(ntr):
    n = int(rng.integers(9, 12)); length = int(rng.choice([150_000, 400_000, 1_000_000]))
    if trial % 2 == 0:
        sites = syn.simulate_sites(n if n % 2 == 0 else n + 1, length, float(rng.choice([0.03, 0.08, 0.15])), seed=int(rng.integers(1, 1 << 30)))
        n = sites.shape[1]
        keys, counts = syn.pattern_table(sites)
    else:
        keys, counts = _copy_mutate_table(rng, n, length, 4)
    names = taxa_names(n)
    dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=int(counts.sum()), taxa=names)
    splits = []
    for _ in range(48):
        k = int(rng.integers(2, n - 1)); left = sorted(rng.choice(n, size=k, replace=False).tolist())
        splits.append((tuple(names[t] for t in left), tuple(names[t] for t in range(n) if t not in left)))
    try:
        got, st = sp.score_splits(dev, splits, return_status=True)
        dn = sp.score_splits(dev, splits, route="dense")
    except Exception as e:
        print("EXC", trial, n, length, len(keys), str(e)[:200]); bad += 1; continue
    for i, spl in enumerate(splits):
        checked += 1
        if abs(got[i] - dn[i]) > 1e-10 or (st[i] & 3):
            bad += 1; print("BAD auto-vs-dense trial", trial, "n", n, "L", length, "D", len(keys), "split", i, [len(spl[0]), len(spl[1])], got[i], dn[i], hex(st[i]))
        if min(len(spl[0]), len(spl[1])) <= 3 and i % 4 == 0:
            M = O.reduced_flattening_packed(keys, counts.astype(np.float64), n, [names.index(t) for t in spl[0]], [names.index(t) for t in spl[1]])[0]
            want = O.dense_split_score(M)
            if abs(want - got[i]) > 1e-10:
                bad += 1; print("BAD oracle trial", trial, "n", n, "D", len(keys), "split", i, M.shape, want, got[i], hex(st[i]))
    print("trial", trial, "n", n, "L", length, "D", len(keys), "its", sorted(set((np.asarray(st) >> 8).tolist()))[:8], "%.0f s" % (time.time() - t0), flush=True)
print("seed", seed0, "trials", ntr, "checked", checked, "bad", bad, "%.0f s" % (time.time() - t0))
