"""Config 3 / 4 sanity + timing (GPU box): 16- and 20-taxon alignments, 1M bp, subflattening route, all splits."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splitp_amd as sp
from splitp_amd import synthetic as syn, batch, _lib
from oracle import splitp_oracle as O

for n, L in ((16, 1_000_000), (20, 1_000_000)):
    t0 = time.time()
    sites = syn.simulate_sites(n, L, 0.05, seed=3)
    keys, counts = syn.pattern_table(sites)
    names = syn.taxa_names(n)
    dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=L, taxa=names)
    splits = list(sp.all_splits(names))
    t1 = time.time()
    taxa_arr, a_arr = batch.encode_splits(splits, dev, n)
    t2 = time.time()
    ctx = dev.ctx
    sc, st = batch.score_encoded(dev, taxa_arr, a_arr, _lib.SP_METHOD_SUBFLATTENING)
    ctx.enable_timing(True); ctx.reset_timing()
    t3 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        sc, st = batch.score_encoded(dev, taxa_arr, a_arr, _lib.SP_METHOD_SUBFLATTENING)
    dt = (time.perf_counter() - t3) / reps
    ph = ctx.phase_times(); ctx.enable_timing(False)
    print(f"n={n} L={L} D={len(keys)} splits={len(splits)} gen {t1-t0:.1f}s encode {t2-t1:.1f}s  score {dt*1e3:.2f} ms -> {len(splits)/dt:.3e} splits/s",
          {k: round(v[0]/max(v[1],1), 3) for k, v in ph.items() if v[1]}, "flagged", int((st & 1).sum()), "sweeps max", int((st >> 8).max()))
    # oracle spot checks (exact moment identity + SciPy SVD)
    M = O.moment_matrix(keys, counts, n)
    for i in (0, len(splits)//3, len(splits)-1):
        oa = taxa_arr[i, :a_arr[i]]; ob = taxa_arr[i, a_arr[i]:]
        S = M[np.ix_(O.subflattening_index(oa, n), O.subflattening_index(ob, n))] / float(L)
        ref = O.dense_split_score(S)
        got_m = sp.subflattening(splits[i], dev)
        assert np.array_equal(np.rint(got_m * L), np.rint(S * L))
        print("   split", i, "ref", ref, "gpu", sc[i], "diff %.1e" % abs(ref - sc[i]))
