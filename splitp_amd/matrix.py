"""Matrix helpers (mirror of reference splitp/matrix.py:4-14)."""
import numpy as np
import scipy.sparse


def is_sparse(matrix):
    """reference: splitp/matrix.py:4-5"""
    return scipy.sparse.issparse(matrix)


def frobenius_norm(matrix, data_table=None):
    """Frobenius norm (reference splitp/matrix.py:7-14, three Python-level loops).

    Same values; computed with one vectorised reduction over the stored entries instead of a
    Python loop with a dok __getitem__ per element (0.1 s per call at nnz = 8113 in the
    reference).  This is host bookkeeping: the scoring path (split_score) never calls it - it gets
    the norm as trace(G) on the device."""
    if data_table is not None:
        return float(np.sqrt(sum(val**2 for _, val in data_table.itertuples(index=False))))
    if is_sparse(matrix):
        coo = matrix.tocoo()
        return float(np.sqrt(np.sum(np.asarray(coo.data, dtype=np.float64) ** 2)))
    return np.sqrt(np.sum(np.asarray(matrix, dtype=np.float64) ** 2))
