"""split_score(scipy sparse matrix) beyond the dense route: big-table form, count-derived and arbitrary values, wide shapes."""
import os, sys, time
import numpy as np, scipy.sparse as sps, scipy.sparse.linalg
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splitp_amd as sp
rng = np.random.default_rng(5)
for kind, (r, c, nnz) in (("counts", (2048, 1500, 150_000)), ("floats", (2048, 1500, 150_000)), ("counts", (70_000, 50_000, 200_000))):
    ri = rng.integers(0, r, nnz); ci = rng.integers(0, c, nnz)
    _, first = np.unique(ri.astype(np.int64) * c + ci, return_index=True)   # distinct cells
    ri, ci = ri[first], ci[first]
    v = rng.integers(1, 50, len(ri)).astype(np.float64)
    # a few heavy rank-one-ish cells so that the spectrum has a top
    v[: 2000] *= 200.0
    v = v / v.sum() if kind == "counts" else v * rng.random(len(ri))
    M = sps.coo_matrix((v, (ri, ci)), shape=(r, c)).tocsr()
    t0 = time.time(); got = sp.split_score(M); dt = time.time() - t0
    top = scipy.sparse.linalg.svds(M, 4, return_singular_vectors=False, tol=0)
    want = float(np.sqrt(max(0.0, 1 - (top ** 2).sum() / (M.data ** 2).sum())))
    print(kind, (r, c), "nnz", len(ri), "want", want, "got", got, "diff %.1e" % abs(want - got), "%.2f s" % dt)
