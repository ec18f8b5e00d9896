"""Randomised hunt on the mutual-information score (fused LDS kernel, global-memory form, float-weight tables)."""
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
for trial in range(ntr):
    n = int(rng.integers(3, 15)); length = int(rng.choice([5, 60, 400, 2500, 20000, 60000])); letters = int(rng.choice([2, 3, 4, 4]))
    keys, counts = _copy_mutate_table(rng, n, length, letters)
    total = int(counts.sum())
    probs = counts / float(total)
    names = taxa_names(n)
    splits = []
    for _ in range(12):
        k = int(rng.integers(1, n)); left = sorted(rng.choice(n, size=k, replace=False).tolist())
        splits.append((tuple(names[t] for t in left), tuple(names[t] for t in range(n) if t not in left)))
    dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=total, taxa=names)
    dev_w = sp.DeviceAlignment.from_arrays(keys, probs, n, taxa=names, exact=False)
    res = {"fused": sp.score_splits(dev, splits, method=sp.Method.mutual_information)}
    sp.get_context().set_option("divergence_global", 1)
    res["global"] = sp.score_splits(dev, splits, method=sp.Method.mutual_information)
    sp.get_context().set_option("divergence_global", 0)
    res["float"] = sp.score_splits(dev_w, splits, method=sp.Method.mutual_information)
    for i, spl in enumerate(splits):
        want = O.rank1_divergence_packed(keys, probs, n, [names.index(t) for t in spl[0]], [names.index(t) for t in spl[1]])
        checked += 1
        for name, arr in res.items():
            if not (abs(arr[i] - want) <= 1e-10 + 1e-12 * abs(want)):
                bad += 1; print("BAD", name, "trial", trial, "n", n, "L", length, "D", len(keys), "split", i, want, arr[i])
print("seed", seed0, "trials", ntr, "checked", checked, "bad", bad, "%.0f s" % (time.time() - t0))
