"""Randomised hunt on split_score(matrix): arbitrary dense / scipy-sparse matrices against numpy's SVD."""
import os, sys, time
import numpy as np
import scipy.sparse as sps
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splitp_amd as sp
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ntr = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(seed0)
def want_score(m):
    if min(m.shape) <= 4:
        return 0.0 if (m != 0).any() else float("nan")
    s = np.linalg.svd(m, compute_uv=False) ** 2
    tot = s.sum()
    if tot == 0: return float("nan")
    x = 1.0 - s[:4].sum() / tot
    return float(np.sqrt(max(x, 0.0)))
def close(a, b):
    if np.isnan(a) or np.isnan(b): return np.isnan(a) and np.isnan(b)
    return abs(a - b) <= 1e-10 or abs(a * a - b * b) <= 5e-14   # (fp64 Gram floor of 1 - top4/trace)
bad = 0; t0 = time.time()
for trial in range(ntr):
    r = int(rng.choice([1, 2, 4, 5, 6, 9, 16, 17, 33, 64, 65, 130, 300, 700, 1024]))
    c = int(rng.choice([1, 3, 4, 5, 7, 16, 40, 64, 257, 900, 2000]))
    kind = int(rng.integers(0, 7))
    if kind == 0:
        m = rng.standard_normal((r, c))
    elif kind == 1:   # low rank + noise
        k = int(rng.integers(1, 9)); m = rng.standard_normal((r, k)) @ rng.standard_normal((k, c)) + 10.0 ** rng.integers(-12, -1) * rng.standard_normal((r, c))
    elif kind == 2:   # sparse counts
        m = np.where(rng.random((r, c)) < rng.choice([0.01, 0.1, 0.5]), rng.integers(1, 1000, (r, c)), 0).astype(np.float64)
    elif kind == 3:   # geometric spectrum
        q = min(r, c); u, _ = np.linalg.qr(rng.standard_normal((r, q))); v, _ = np.linalg.qr(rng.standard_normal((c, q)))
        m = (u * (float(rng.choice([0.3, 0.7, 0.95, 0.999])) ** np.arange(q))) @ v.T
    elif kind == 4:   # clustered: many equal singular values
        q = min(r, c); u, _ = np.linalg.qr(rng.standard_normal((r, q))); v, _ = np.linalg.qr(rng.standard_normal((c, q)))
        sv = np.ones(q); sv[: min(q, int(rng.integers(0, 4)))] = 5.0; m = (u * sv) @ v.T
    elif kind == 5:   # probability-like
        m = rng.random((r, c)) ** 8; m /= m.sum()
    else:             # all zero / single entry
        m = np.zeros((r, c)); 
        if rng.random() < 0.5: m[rng.integers(r), rng.integers(c)] = 3.0
    scale = float(rng.choice([-8, -3, 0, 0, 0, 5]))
    m = m * 10.0 ** scale
    w = want_score(m)
    try:
        g = float(sp.split_score(m))
        g2 = float(sp.split_score(sps.csr_matrix(m))) if (kind in (2, 6) or trial % 5 == 0) else g
    except Exception as e:
        print("EXC", trial, (r, c), kind, str(e)[:160]); bad += 1; continue
    for name, val in (("dense", g), ("coo", g2)):
        if not close(w, val):
            bad += 1; print("BAD", name, "trial", trial, (r, c), "kind", kind, "scale", scale, "want", w, "got", val)
            if bad < 12 and os.path.isdir("gpurun_out"): np.save("gpurun_out/badm_%d_%d.npy" % (seed0, trial), m)
print("seed", seed0, "trials", ntr, "bad", bad, "%.0f s" % (time.time() - t0))
