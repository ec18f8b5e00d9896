import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splitp_amd as sp
from oracle import splitp_oracle as O
from tests.conftest import taxa_names
from tests.test_gpu_parity import _copy_mutate_table
rng = np.random.default_rng(77)
for trial in range(10):
    n = int(rng.integers(5, 14)); length = int(rng.choice([20, 150, 900, 6000])); letters = int(rng.choice([2, 3, 4, 4]))
    keys, counts = _copy_mutate_table(rng, n, length, letters)
    names = taxa_names(n)
    if n <= 8:
        splits = list(sp.all_splits(names))
    else:
        for _ in range(24):
            k = int(rng.integers(2, n - 1)); left = sorted(rng.choice(n, size=k, replace=False).tolist())
w = counts / float(counts.sum())
spl = splits[16]
oa = [names.index(t) for t in spl[0]]; ob = [names.index(t) for t in spl[1]]
M = O.reduced_flattening_packed(keys, w, n, oa, ob)[0]
print("shape", M.shape, "oracle", O.dense_split_score(M), "split_score(M)", sp.split_score(M), "split_score(M.T)", sp.split_score(M.T.copy()))
dev_w = sp.DeviceAlignment.from_arrays(keys, w, n, taxa=names, exact=False)
for sub in ([spl], splits[14:18], splits):
    g, s = sp.score_splits(dev_w, sub, return_status=True)
    idx = sub.index(spl)
    print("batch of", len(sub), "->", g[idx], hex(s[idx]))
dev_w.ctx.set_gram_mode("f64")
dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=int(counts.sum()), taxa=names)
g, s = sp.score_splits(dev, splits, return_status=True, route="dense")
print("exact table, f64 gram, dense:", g[16], hex(s[16]))
for scale in (1.0, 1e3, 1e6, 6000.0):
    d2 = sp.DeviceAlignment.from_arrays(keys, w * scale, n, taxa=names, exact=False)
    g, s = sp.score_splits(d2, [spl], return_status=True)
    print("scale", scale, "->", g[0], hex(s[0]))
# other 3|3 splits of this table
g, s = sp.score_splits(dev_w, splits, return_status=True)
gd = sp.score_splits(dev, splits, route="dense")
print("wrong ones:", [(i, len(splits[i][0]), g[i], gd[i]) for i in range(len(splits)) if abs(g[i] - gd[i]) > 1e-9])
print("---- scale bisect")
for scale in (0.3, 1.0, 2.0, 4.0, 8.0, 16.0, 32.0, 64.0, 128.0, 256.0):
    d2 = sp.DeviceAlignment.from_arrays(keys, w * scale, n, taxa=names, exact=False)
    g, s = sp.score_splits(d2, [spl], return_status=True)
    print("scale", scale, "->", g[0], hex(s[0]))
Mw = O.reduced_flattening_packed(keys, w, n, oa, ob)[0]
G = Mw.T @ Mw if Mw.shape[1] < Mw.shape[0] else Mw @ Mw.T
ev = np.linalg.eigvalsh(G)[::-1]
print("G shape", G.shape, "eig", ev[:8], "min", ev[-3:], "trace", np.trace(G))
print("---- matrix path at scales")
for scale in (0.3, 1.0, 4.0, 64.0, 256.0, 1000.0):
    print(scale, sp.split_score(Mw * scale), sp.split_score((Mw * scale).T.copy()))
# table path, transposed split (cols <-> rows)
splT = (spl[1], spl[0])
print("table path swapped split:", sp.score_splits(dev_w, [splT], return_status=True))
print("table path, sparse format score:", sp.split_score(sp.flattening(spl, dict(dev_w.items()))))
