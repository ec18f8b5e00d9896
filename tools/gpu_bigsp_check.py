import sys, numpy as np, scipy.sparse as sps, time
sys.path.insert(0, '/root/repo' if len(sys.argv) < 2 else sys.argv[1])
import splitp_amd as sp
rng = np.random.default_rng(5)
for kind in ("counts", "floats"):
    r, c, nnz = 2048, 1500, 150_000
    ri = rng.integers(0, r, nnz); ci = rng.integers(0, c, nnz)
    _, first = np.unique(ri.astype(np.int64) * c + ci, return_index=True)   # distinct cells
    ri, ci = ri[first], ci[first]
    v = rng.integers(1, 50, len(ri)).astype(np.float64)
    v = v / v.sum() if kind == "counts" else rng.random(len(ri)) ** 3
    # low-rank-ish structure so that the spectrum has a gap
    M = sps.coo_matrix((v, (ri, ci)), shape=(r, c))
    t0 = time.time(); got = sp.split_score(M.tocsr()); dt = time.time() - t0
    s2 = np.linalg.svd(M.toarray(), compute_uv=False) ** 2
    want = float(np.sqrt(max(0.0, 1 - s2[:4].sum() / s2.sum())))
    print(kind, "nnz", len(ri), "want", want, "got", got, "diff %.1e" % abs(want - got), "%.2f s" % dt)
