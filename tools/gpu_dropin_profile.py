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
from splitp_amd import constructions as K
ref = np.array(sp.score_splits(table, splits))
for prefetch in (False, True, False, True):
    K.PREFETCH_SCORES = prefetch
    loop()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); got = loop(); best = min(best, time.perf_counter() - t0)
    assert np.array_equal(np.array(got), ref)
    print("prefetch %d: loop %.3f s = %.0f splits/s (best of 3, scores bit-equal to the batched call)"
          % (prefetch, best, len(splits) / best))
pr = cProfile.Profile(); pr.enable(); loop(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:4500])
