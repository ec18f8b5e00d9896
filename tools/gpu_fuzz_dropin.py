"""Randomised hunt on the drop-in loop with the score prefetch (GPU box; not part of the suite):
    python tools/gpu_fuzz_dropin.py SEED TRIALS
Random tables (4 - 11 taxa, counts / float weights / plain dicts / DeviceAlignments), two tables interleaved, random splits;
per step one of: score the flattening at once (the README loop), keep it for later, drop it unscored, edit it in place, ask
twice.  Every score of an untouched flattening must equal the batched call's bit for bit (same kernels, same resident table),
every edited one the oracle's score of the edited matrix."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splitp_amd as sp
from splitp_amd import constructions as K, device
from oracle import splitp_oracle as O
from tests.conftest import taxa_names
from tests.test_gpu_parity import _copy_mutate_table

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ntr = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rng = np.random.default_rng(seed0)
bad = 0; checked = 0; prefetched = 0; overtaken = 0; edited = 0; t0 = time.time()


def make_table(trial):
    n = int(rng.integers(4, 12)); length = int(rng.choice([60, 400, 2500, 20000])); letters = int(rng.choice([2, 3, 4, 4]))
    keys, counts = _copy_mutate_table(rng, n, length, letters)
    names = taxa_names(n)
    kind = int(rng.integers(0, 4))
    if kind == 0:       # resident count table
        tab = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=int(counts.sum()), taxa=names)
    elif kind == 1:     # resident float-weight table (dense route)
        w = counts / float(counts.sum()) * rng.uniform(0.5, 1.5, size=len(counts))
        tab = sp.DeviceAlignment.from_arrays(keys, w, n, taxa=names, exact=False)
    else:               # the reference's plain dict (count-derived probabilities, or perturbed ones)
        vals = counts / float(counts.sum())
        if kind == 3:
            vals = vals * rng.uniform(0.5, 1.5, size=len(vals))
        tab = O.unpack_table(keys, vals, n)
    splits = []
    for _ in range(int(rng.integers(6, 20))):
        k = int(rng.integers(2, n - 1)); left = sorted(rng.choice(n, size=k, replace=False).tolist())
        splits.append((tuple(names[t] for t in left), tuple(names[t] for t in range(n) if t not in left)))
    return tab, splits


def same(a, b):
    return np.float64(a).view(np.uint64) == np.float64(b).view(np.uint64)


warnings.simplefilter("ignore", RuntimeWarning)
def one_trial(trial):
    global bad, checked, prefetched, overtaken, edited
    tabs = [make_table(trial) for _ in range(int(rng.integers(1, 3)))]
    want = []
    K.PREFETCH_SCORES = False
    for tab, splits in tabs:
        # the synchronous call, one split at a time (the prefetch must reproduce it bit for bit); the batched call agrees
        # within the tolerance (on the dense route a batch may tile differently)
        one = np.array([sp.split_score(sp.flattening(spl, tab, sp.FlatFormat.reduced)) for spl in splits])
        many = np.asarray(sp.score_splits(tab, splits))
        if not np.all((np.abs(one - many) <= 1e-10) | (np.abs(one * one - many * many) <= 8e-15)):
            bad += 1; print("BAD batched-vs-single", trial, float(np.max(np.abs(one - many))))
        want.append(one)
    kept = []       # (F, table index, split index)
    K.PREFETCH_SCORES = bool(rng.integers(0, 4))          # (a quarter of the trials without the prefetch: same numbers)
    hoard = int(rng.integers(0, 4)) == 0                  # a quarter of the trials keep many matrices alive: ring slots overtaken
    for step in range(int(rng.integers(150, 400)) if hoard else int(rng.integers(10, 90))):
        ti = int(rng.integers(0, len(tabs))); tab, splits = tabs[ti]
        si = int(rng.integers(0, len(splits)))
        F = sp.flattening(splits[si], tab, sp.FlatFormat.reduced)
        prefetched += F._sp_pending is not None
        act = int(rng.integers(0, 10))
        if F.size == 0 or min(F.shape) < 1:
            continue
        if act <= 4:                                    # the README loop
            s = sp.split_score(F); checked += 1
            if not same(s, want[ti][si]):
                bad += 1; print("BAD now", trial, step, ti, si, F.shape, s, want[ti][si])
            if act == 4 and not same(sp.split_score(F), want[ti][si]):
                bad += 1; print("BAD twice", trial, step, ti, si)
        elif act <= 6 or (hoard and act == 7):
            kept.append((F, ti, si))
        elif act == 7:
            pass                                        # dropped unscored
        else:                                           # edited in place: the matrix it now is
            if min(F.shape) > 4 and max(F.shape) <= 1024 and F.size <= 300_000:
                i, j = int(rng.integers(0, F.shape[0])), int(rng.integers(0, F.shape[1]))
                F[i, j] += float(np.max(F)) * 0.5 + 1e-3
                s = sp.split_score(F); checked += 1; edited += 1
                ref = O.dense_split_score(np.asarray(F))
                if not (abs(s - ref) <= 1e-10 or abs(s * s - ref * ref) <= 8e-15):
                    bad += 1; print("BAD edited", trial, step, ti, si, F.shape, s, ref)
        if kept and rng.integers(0, 150 if hoard else 5) == 0:
            rng.shuffle(kept)
            for F2, t2, s2 in kept:
                had = F2._sp_pending
                if had is not None and had[0].ticket[had[1] % K._PREFETCH_SLOTS] != had[1]:
                    overtaken += 1
                s = sp.split_score(F2); checked += 1
                if not same(s, want[t2][s2]):
                    bad += 1; print("BAD kept", trial, step, t2, s2, F2.shape, s, want[t2][s2])
            kept = []
    for F2, t2, s2 in kept:
        s = sp.split_score(F2); checked += 1
        if not same(s, want[t2][s2]):
            bad += 1; print("BAD kept-end", trial, t2, s2, F2.shape, s, want[t2][s2])
    K.PREFETCH_SCORES = True
    device.clear_table_cache()
for trial in range(ntr):
    try:
        one_trial(trial)
    except Exception as e:      # noqa: BLE001
        bad += 1; print("EXC", trial, type(e).__name__, str(e)[:300])
        K.PREFETCH_SCORES = True
        device.clear_table_cache()
print("seed", seed0, "trials", ntr, "scores checked", checked, "bad", bad, "flattenings prefetched", prefetched, "overtaken slots", overtaken,
      "edited in place", edited, "%.0f s" % (time.time() - t0))
