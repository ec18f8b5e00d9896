"""cProfile of the unchanged README loop on the drop-in functions (GPU box): where the 0.3 - 0.45 s per 501 splits go."""
import cProfile, pstats, sys, time, io
sys.path.insert(0, '.')
import numpy as np
import splitp_amd as sp
from splitp_amd import synthetic as syn
n, L = 10, 100_000
names = syn.taxa_names(n)
keys, counts = syn.pattern_table(syn.simulate_sites(n, L, 0.05, seed=1))
table = syn.table_as_dict(keys, counts, n, total=L)
splits = list(sp.all_splits(names))
def loop():
    out = []
    for split in splits:
        F = sp.flattening(split, table, sp.FlatFormat.reduced)
        out.append(sp.split_score(F))
    return out
loop()
t0 = time.perf_counter(); loop(); dt = time.perf_counter() - t0
print("loop: %.3f s = %.0f splits/s" % (dt, len(splits) / dt))
pr = cProfile.Profile(); pr.enable(); loop(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:4500])
