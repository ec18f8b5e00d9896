"""Histogram of the half products (status >> 8) the sparse route needs on the benchmark table, by size class."""
import sys, collections
import numpy as np
sys.path.insert(0, '.')
import splitp_amd as sp
from splitp_amd import synthetic as syn, batch, _lib
n, L = 10, 100_000
names = syn.taxa_names(n)
keys, counts = syn.pattern_table(syn.simulate_sites(n, L, 0.05, seed=1))
dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=L, taxa=names)
taxa_arr, a_arr = sp.encode_all_splits(n)
sc, st = batch.score_encoded(dev, taxa_arr, a_arr, _lib.SP_METHOD_FLATTENING)
k = np.minimum(a_arr, n - a_arr)
for kk in (2, 3, 4, 5):
    it = st[k == kk] >> 8
    print("k", kk, "splits", int((k == kk).sum()), "iterations:", dict(sorted(collections.Counter(it.tolist()).items())),
          "scores %.4f..%.4f" % (sc[k == kk].min(), sc[k == kk].max()))
