"""Two-splits-a-wave subflattening kernel (subflat_pair.hip) against the one-split-a-wave kernel (option subscore_pair = 0)
on the same batches (GPU box):  python tools/gpu_subpair_check.py
All splits (enumerated on the device) and shuffled lists of 6 - 20 taxon tables, count and float-weight tables, a
degenerate table; prints the largest differences in score and in score^2."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import splitp_amd as sp
from splitp_amd import synthetic as syn

ctx = sp.get_context()
worst = 0.0
for n, length, seed in ((6, 5000, 1), (8, 20_000, 2), (10, 50_000, 3), (12, 80_000, 7), (14, 100_000, 4), (16, 300_000, 5), (20, 300_000, 6)):
    sites = syn.simulate_sites(n, length, 0.05, seed=seed)
    keys, counts = syn.pattern_table(sites)
    names = syn.taxa_names(n)
    tables = [("counts", sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=length, taxa=names)),
              ("float", sp.DeviceAlignment.from_arrays(keys, counts / float(length), n, taxa=names, exact=False)),
              ("degenerate", sp.DeviceAlignment.from_arrays(keys[:3], None, n, counts=counts[:3], n_sites=int(counts[:3].sum()), taxa=names))]
    for label, dev in tables:
        for trivial in (False, True):
            res = {}
            for pair in (1, 0):
                ctx.set_option("subscore_pair", pair)
                res[pair], st = sp.score_all_splits(dev, method=sp.Method.subflattening, trivial=trivial, return_status=True)
                if pair == 1:
                    flagged = int(np.count_nonzero(st & 3))
                    ps = (st >> 8)[st != 0]
                    passes = "passes mean %.2f max %d" % (ps.mean(), ps.max()) if ps.size else "passes -"
            ctx.set_option("subscore_pair", 1)
            a, b = res[1], res[0]
            nan = np.isnan(a) & np.isnan(b)
            d = np.where(nan, 0.0, np.abs(a - b))
            d2 = np.where(nan, 0.0, np.abs(a * a - b * b))
            bad = ~((d <= 1e-11) | (d2 <= 1e-13)) | (np.isnan(a) != np.isnan(b))
            print(f"n {n:2d} {label:10s} trivial {int(trivial)}: {a.size:7d} splits  max |d| {np.nanmax(d):.2e}  max |d2| {np.nanmax(d2):.2e}  bad {int(bad.sum())}  {passes}  flagged {flagged}")
            worst = max(worst, float(np.nanmax(np.minimum(d, d2 * 100))))
            if flagged:
                sys.exit(1)
            if bad.any():
                i = int(np.nonzero(bad)[0][0])
                print("   first bad", i, a[i], b[i])
                sys.exit(1)
        if n <= 14 and label == "counts":
            allsp = list(sp.all_splits(names))
            rng = np.random.default_rng(seed)
            pick = [allsp[i] for i in rng.permutation(len(allsp))[:999]]
            ctx.set_option("subscore_pair", 1)
            a = sp.score_splits(dev, pick, method=sp.Method.subflattening)
            ctx.set_option("subscore_pair", 0)
            b = sp.score_splits(dev, pick, method=sp.Method.subflattening)
            ctx.set_option("subscore_pair", 1)
            full = dict(zip(allsp, sp.score_all_splits(dev, method=sp.Method.subflattening)))
            same = np.array([full[s] for s in pick])
            print(f"n {n:2d} shuffled list of {len(pick)}: max |pair - single| {np.max(np.abs(a - b)):.2e}; list == all-splits bits: {np.array_equal(a, same)}")
            if not np.array_equal(a, same) or np.max(np.abs(a - b)) > 1e-11:
                sys.exit(1)
print("OK")
