"""split_score - drop-in for reference splitp/phylogenetics.py:280-328, on MI355X."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .device import get_context
from .matrix import is_sparse


def split_score(matrix, return_singular_values=False, force_frob_norm_on_dense=False, data_table_for_frob_norm=None):
    """score = sqrt(1 - (sum of the 4 largest sigma^2) / (sum of all sigma^2)).

    reference: splitp/phylogenetics.py:315-328 dispatching to :280-300 (dense, LAPACK gesdd) or
    :303-312 (sparse, ARPACK svds + Frobenius norm).  Both run here through the same device route:
    Gram matrix over the smaller side by fp64 MFMA, top-4 eigenvalues by block subspace iteration,
    trace for the denominator.  The three optional arguments are accepted for signature
    compatibility; in the reference the two booleans are misrouted no-ops and
    `data_table_for_frob_norm` only changes how the same norm is summed.

    Returns np.float64 (dense) / float (sparse) like the reference.  Where the reference's dense
    path can return nan from a slightly negative operand (no clamp, :293-300) this returns 0.0,
    like the reference's own sparse path (:311-312)."""
    ctx = get_context()
    lib = ctx._lib
    out = C.c_double()
    if is_sparse(matrix):
        coo = matrix.tocoo()
        ri = np.ascontiguousarray(coo.row, dtype=np.int64)
        ci = np.ascontiguousarray(coo.col, dtype=np.int64)
        v = np.ascontiguousarray(coo.data, dtype=np.float64)
        _lib.check(lib.sp_score_coo_f64(ctx.handle, _lib._ptr(ri, C.c_int64), _lib._ptr(ci, C.c_int64),
                                        _lib._ptr(v, C.c_double), len(v), matrix.shape[0], matrix.shape[1],
                                        C.byref(out)))
        return float(out.value)
    m = np.ascontiguousarray(np.array(matrix), dtype=np.float64)
    if m.ndim != 2:
        raise ValueError("split_score expects a 2-D matrix")
    _lib.check(lib.sp_score_matrix_f64(ctx.handle, _lib._ptr(m, C.c_double), m.shape[0], m.shape[1], m.shape[1],
                                       C.byref(out)))
    return np.float64(out.value)


def flattening_rank_1_approximation_divergence(flattening):
    """Kullback-Leibler divergence of a flattening from its rank-1 approximation (the product of its marginals):
    sum over the non-zero cells of f * log(f / (column sum * row sum)).

    reference: splitp/phylogenetics.py:364-373 with the marginals of :332-341; it is the score erickson_SVD uses
    with Method.mutual_information (:135-140).  Computed on the device (row sums, column sums and the row-ordered
    sum, `csrc/divergence.hip`); returns np.float64 like the reference's numpy accumulation."""
    ctx = get_context()
    m = np.ascontiguousarray(np.array(flattening), dtype=np.float64)
    if m.ndim != 2:
        raise ValueError("flattening_rank_1_approximation_divergence expects a 2-D matrix")
    out = C.c_double()
    _lib.check(ctx._lib.sp_divergence_matrix_f64(ctx.handle, _lib._ptr(m, C.c_double), m.shape[0], m.shape[1],
                                                 m.shape[1], C.byref(out)))
    return np.float64(out.value)
