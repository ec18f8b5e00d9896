import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splitp_amd as sp
from oracle import splitp_oracle as O
from tests.conftest import taxa_names
from tests.test_gpu_parity import _copy_mutate_table
rng = np.random.default_rng(77)
for trial in range(30):
    n = int(rng.integers(5, 14)); length = int(rng.choice([20, 150, 900, 6000])); letters = int(rng.choice([2, 3, 4, 4]))
    keys, counts = _copy_mutate_table(rng, n, length, letters)
    names = taxa_names(n)
    if n <= 8:
        splits = list(sp.all_splits(names))
    else:
        splits = []
        for _ in range(24):
            k = int(rng.integers(2, n - 1)); left = sorted(rng.choice(n, size=k, replace=False).tolist())
            splits.append((tuple(names[t] for t in left), tuple(names[t] for t in range(n) if t not in left)))
    if not (n <= 10 and trial % 3 == 0): continue
    dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=int(counts.sum()), taxa=names)
    got, st = sp.score_splits(dev, splits, return_status=True)
    dev_w = sp.DeviceAlignment.from_arrays(keys, counts / float(counts.sum()), n, taxa=names, exact=False)
    gw, sw = sp.score_splits(dev_w, splits, return_status=True)
    bad = [i for i in range(len(splits)) if not (abs(gw[i] - got[i]) <= 1e-10)]
    print("trial", trial, "n", n, "L", length, "letters", letters, "D", len(keys), "bad", len(bad))
    for i in bad[:4]:
        spl = splits[i]
        M = O.reduced_flattening_packed(keys, counts.astype(np.float64), n, [names.index(t) for t in spl[0]], [names.index(t) for t in spl[1]])[0]
        want = 0.0 if min(M.shape) <= 4 else O.dense_split_score(M)
        print("   ", i, len(spl[0]), M.shape, "oracle", want, "sparse", got[i], hex(st[i]), "float", gw[i], hex(sw[i]))
