"""Randomised hunt on the batched subflattening score (Householder + Sturm kernel) against the oracle's moment identity."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splitp_amd as sp
from oracle import splitp_oracle as O
from tests.conftest import taxa_names
from tests.test_gpu_parity import _copy_mutate_table
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ntr = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(seed0)
bad = 0; checked = 0; t0 = time.time()
def close(a, b):
    if np.isnan(a) or np.isnan(b): return np.isnan(a) and np.isnan(b)
    return abs(a - b) <= 1e-10 or abs(a * a - b * b) <= 5e-14
for trial in range(ntr):
    n = int(rng.integers(3, 21)); length = int(rng.choice([3, 10, 60, 400, 2500, 20000])); letters = int(rng.choice([1, 2, 3, 4, 4]))
    keys, counts = _copy_mutate_table(rng, n, length, max(letters, 1))
    if letters == 1: keys = keys[:1]; counts = counts[:1]
    if trial % 7 == 0: counts = counts * int(rng.choice([300, 70_000]))
    names = taxa_names(n)
    total = int(counts.sum())
    exact = trial % 3 != 0
    dev = (sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=total, taxa=names) if exact else
           sp.DeviceAlignment.from_arrays(keys, counts / float(total), n, taxa=names, exact=False))
    splits = []
    for _ in range(24):
        k = int(rng.integers(1, n)); left = sorted(rng.choice(n, size=k, replace=False).tolist())
        splits.append((tuple(names[t] for t in left), tuple(names[t] for t in range(n) if t not in left)))
    try:
        got, st = sp.score_splits(dev, splits, method=sp.Method.subflattening, return_status=True)
    except Exception as e:
        print("EXC", trial, n, length, letters, str(e)[:200]); bad += 1; continue
    M = O.moment_matrix(keys, counts, n)
    for i, spl in enumerate(splits):
        oa = [names.index(t) for t in spl[0]]; ob = [names.index(t) for t in spl[1]]
        S = M[np.ix_(O.subflattening_index(oa, n), O.subflattening_index(ob, n))] / float(total)
        want = 0.0 if min(S.shape) <= 4 else O.dense_split_score(S)
        if np.isnan(want) and (S != 0).any(): want = 0.0
        checked += 1
        if not close(want, got[i]) or (st[i] & 3):
            bad += 1
            print("BAD trial", trial, "n", n, "L", length, "letters", letters, "exact", exact, "split", i, S.shape, "want", want, "got", got[i], hex(st[i]))
print("seed", seed0, "trials", ntr, "checked", checked, "bad", bad, "%.0f s" % (time.time() - t0))
