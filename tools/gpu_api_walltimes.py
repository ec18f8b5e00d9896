"""Wall time of the Python entry points a user of the reference would call (GPU box): where the host costs sit."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import splitp_amd as sp
from splitp_amd import synthetic as syn
n, L = 10, 100_000
names = syn.taxa_names(n)
keys, counts = syn.pattern_table(syn.simulate_sites(n, L, 0.05, seed=1))
table = syn.table_as_dict(keys, counts, n, total=L)
def best(f, reps=20):
    f(); t = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
    return min(t) * 1e3, float(np.median(t)) * 1e3
dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=L, taxa=names)
splits = list(sp.all_splits(names))
print("DeviceAlignment.from_table(dict)      min %.3f ms  median %.3f ms" % best(lambda: sp.DeviceAlignment.from_table(table, taxa=names), 5))
print("DeviceAlignment.from_arrays           min %.3f ms  median %.3f ms" % best(lambda: sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=L, taxa=names)))
print("score_all_splits(dev)                 min %.3f ms  median %.3f ms" % best(lambda: sp.score_all_splits(dev)))
print("score_splits(dev, list of 501 splits) min %.3f ms  median %.3f ms" % best(lambda: sp.score_splits(dev, splits)))
print("score_splits(dict, list of 501)       min %.3f ms  median %.3f ms" % best(lambda: sp.score_splits(table, splits), 5))
print("score_all_splits(dev, subflattening)  min %.3f ms  median %.3f ms" % best(lambda: sp.score_all_splits(dev, method=sp.Method.subflattening)))
print("score_all_splits(dev, route=dense)    min %.3f ms  median %.3f ms" % best(lambda: sp.score_all_splits(dev, route="dense"), 5))
from splitp_amd import inference
t0 = time.perf_counter(); tree = inference.erickson_SVD(dev, taxa=names); dt = time.perf_counter() - t0
print("erickson_SVD(dev) (8 rounds)          %.3f ms" % (dt * 1e3))
t0 = time.perf_counter(); tree = inference.erickson_SVD(dev, taxa=names); dt = time.perf_counter() - t0
print("erickson_SVD(dev) again               %.3f ms" % (dt * 1e3))
