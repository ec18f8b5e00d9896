"""Long randomised hunt (not part of the test suite): many seeds of the sweeps in tests/test_gpu_parity.py."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splitp_amd as sp
from oracle import splitp_oracle as O
from tests.conftest import taxa_names
from tests.test_gpu_parity import _copy_mutate_table
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ntr = int(sys.argv[2]) if len(sys.argv) > 2 else 200
nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 12
rng = np.random.default_rng(seed0)
bad = 0; checked = 0; t0 = time.time(); flagged = 0; direct = 0
def close(a, b):
    # score within 1e-10, or 1 - top4/trace within 8e-15 (the fp64 floor of that difference: it decides scores below ~2e-5;
    # 4e-15 until round 2 - tables whose counts are split into 16-bit pieces reach 5.4e-15)
    return abs(a - b) <= 1e-10 or abs(a * a - b * b) <= 8e-15
for trial in range(ntr):
    n = int(rng.integers(4, nmax + 1)); length = int(rng.choice([10, 60, 400, 2500, 20000])); letters = int(rng.choice([2, 3, 4, 4]))
    keys, counts = _copy_mutate_table(rng, n, length, letters)
    if trial % 7 == 0: counts = counts * int(rng.choice([300, 70_000]))
    names = taxa_names(n)
    dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=int(counts.sum()), taxa=names)
    if n <= 7:
        splits = list(sp.all_splits(names))
    else:
        splits = []
        for _ in range(16):
            k = int(rng.integers(2, n - 1)); left = sorted(rng.choice(n, size=k, replace=False).tolist())
            splits.append((tuple(names[t] for t in left), tuple(names[t] for t in range(n) if t not in left)))
    try:
        got, st = sp.score_splits(dev, splits, return_status=True)
    except Exception as e:
        print("EXC", trial, n, length, letters, len(keys), str(e)[:200]); bad += 1; continue
    flagged += int(np.count_nonzero(st & 3)); direct += int(np.count_nonzero(st & 4))
    dn = None
    if n <= 10:
        dn = sp.score_splits(dev, splits, route="dense")
        dw = sp.score_splits(sp.DeviceAlignment.from_arrays(keys, counts / float(counts.sum()), n, taxa=names, exact=False), splits)
    for i, spl in enumerate(splits):
        M = O.reduced_flattening_packed(keys, counts.astype(np.float64), n, [names.index(t) for t in spl[0]], [names.index(t) for t in spl[1]])[0]
        if min(M.shape) > 400: continue
        want = 0.0 if min(M.shape) <= 4 else O.dense_split_score(M)
        if np.isnan(want): want = 0.0
        checked += 1
        for name, val in (("sparse", got[i]), ("dense", dn[i] if dn is not None else want), ("float", dw[i] if dn is not None else want)):
            if not close(want, val):
                bad += 1
                print("BAD", name, "trial", trial, "n", n, "L", length, "letters", letters, "D", len(keys), "split", i, M.shape, "want", want, "got", val, hex(st[i]))
print("seed", seed0, "trials", ntr, "checked", checked, "bad", bad, "flagged(status bit 0/1)", flagged, "finished by the direct solver", direct,
      "%.0f s" % (time.time() - t0))
