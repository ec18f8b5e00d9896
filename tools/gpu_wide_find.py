"""Find seeds whose 12-taxon copy-mutate tables send splits to the wide fallback block (status its > 41)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splitp_amd as sp
from oracle import splitp_oracle as O
from tests.conftest import taxa_names
from tests.test_gpu_parity import _copy_mutate_table
for seed in range(12):
    rng = np.random.default_rng(seed)
    n = 12
    length = [2500, 20000][seed % 2]; letters = 3
    keys, counts = _copy_mutate_table(rng, n, length, letters)
    names = taxa_names(n)
    dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=int(counts.sum()), taxa=names)
    splits = []
    for _ in range(16):
        k = int(rng.integers(2, n - 1)); left = sorted(rng.choice(n, size=k, replace=False).tolist())
        splits.append((tuple(names[t] for t in left), tuple(names[t] for t in range(n) if t not in left)))
    t0 = time.time()
    got, st = sp.score_splits(dev, splits, return_status=True)
    dt = time.time() - t0
    worst = 0.0; nchk = 0
    for i, spl in enumerate(splits):
        M = O.reduced_flattening_packed(keys, counts.astype(np.float64), n, [names.index(t) for t in spl[0]], [names.index(t) for t in spl[1]])[0]
        if min(M.shape) > 400: continue
        want = 0.0 if min(M.shape) <= 4 else O.dense_split_score(M)
        worst = max(worst, abs(want - got[i])); nchk += 1
    print("seed", seed, "L", length, "D", len(keys), "its", sorted(set((np.asarray(st) >> 8).tolist())), "flags", sorted(set((np.asarray(st) & 3).tolist())), "checked", nchk, "worst %.2e" % worst, "%.3f s" % dt)
