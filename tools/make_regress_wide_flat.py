"""Regression fixture for the wide block's false acceptance: the two tables the second soak of round 3
found (tools/gpu_fuzz_long.py seed 71002 trial 469, nmax 14; seed 71004 trial 665, nmax 12; profiles/r03_fuzz_soak_final_build_2.txt)
are regenerated on the CPU by replaying the tool's NumPy random stream, and stored with the oracle's scores of their 16 splits:
    python tools/make_regress_wide_flat.py   ->  tests/golden/regress_wide_flat_spectrum.npz
No GPU and no reference import needed (the oracle is pinned by tests/test_oracle_golden.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import splitp_oracle as O  # noqa: E402
from tests.test_gpu_parity import _copy_mutate_table  # noqa: E402


def replay(seed0, nmax, want_trial):
    rng = np.random.default_rng(seed0)
    for trial in range(want_trial + 1):
        n = int(rng.integers(4, nmax + 1))
        length = int(rng.choice([10, 60, 400, 2500, 20000]))
        letters = int(rng.choice([2, 3, 4, 4]))
        keys, counts = _copy_mutate_table(rng, n, length, letters)
        if trial % 7 == 0:
            counts = counts * int(rng.choice([300, 70_000]))
        lefts = []
        if n > 7:
            for _ in range(16):
                k = int(rng.integers(2, n - 1))
                lefts.append(sorted(rng.choice(n, size=k, replace=False).tolist()))
    return n, length, letters, keys, counts, lefts


out = {}
for tag, seed0, nmax, trial in (("a", 71002, 14, 469), ("b", 71004, 12, 665)):
    n, length, letters, keys, counts, lefts = replay(seed0, nmax, trial)
    print(tag, "seed", seed0, "trial", trial, "n", n, "L", length, "letters", letters, "D", len(keys))
    want = []
    left_arr = np.full((len(lefts), n), -1, dtype=np.int32)
    for i, left in enumerate(lefts):
        right = [t for t in range(n) if t not in left]
        m = O.reduced_flattening_packed(keys, counts.astype(np.float64), n, left, right)[0]
        s = 0.0 if min(m.shape) <= 4 else float(O.dense_split_score(m))
        want.append(s)
        left_arr[i, : len(left)] = left
        print("   split", i, left, m.shape, s)
    out[tag + "_n"] = np.int32(n)
    out[tag + "_keys"] = keys
    out[tag + "_counts"] = counts
    out[tag + "_left"] = left_arr
    out[tag + "_want"] = np.array(want)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "regress_wide_flat_spectrum.npz"), **out)
print("written")
