"""split_score - drop-in for reference splitp/phylogenetics.py:280-328, on MI355X."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .constructions import flattening_origin
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
    origin = flattening_origin(matrix)
    if origin is not None:
        # an untouched flattening(..., FlatFormat.reduced) of a resident table: score the split where the table lives
        from .batch import score_encoded
        from .constructions import take_prefetched_score

        al, oa, ob = origin
        ready = take_prefetched_score(matrix, al)      # enqueued by flattening() behind the fetch (constructions.py)
        if ready is not None:
            return ready
        al.ctx.sync_stream_with_torch()
        taxa_arr = np.ascontiguousarray(np.concatenate([oa, ob])[None, :], dtype=np.int32)
        scores, _ = score_encoded(al, taxa_arr, np.array([len(oa)], dtype=np.int32), _lib.SP_METHOD_FLATTENING)
        return np.float64(scores[0])
    ctx = get_context()
    lib = ctx._lib
    out = C.c_double()
    if is_sparse(matrix):
        coo = matrix.tocoo()
        ri = np.ascontiguousarray(coo.row, dtype=np.int64)
        ci = np.ascontiguousarray(coo.col, dtype=np.int64)
        v = np.ascontiguousarray(coo.data, dtype=np.float64)
        big = _score_big_sparse(ri, ci, v, matrix.shape)
        if big is not None:
            return big
        _lib.check(lib.sp_score_coo_f64(ctx.handle, _lib._ptr(ri, C.c_int64), _lib._ptr(ci, C.c_int64),
                                        _lib._ptr(v, C.c_double), len(v), matrix.shape[0], matrix.shape[1],
                                        C.byref(out)))
        return float(out.value)
    m = np.ascontiguousarray(np.array(matrix), dtype=np.float64)
    if m.ndim != 2:
        raise ValueError("split_score expects a 2-D matrix")
    if min(m.shape) > 1024:      # a reduced flattening of 12+ taxa: beyond the dense route, try the sparse kernel
        nz_r, nz_c = np.nonzero(m)
        big = _score_big_sparse(nz_r.astype(np.int64), nz_c.astype(np.int64), m[nz_r, nz_c], m.shape)
        if big is not None:
            return np.float64(big)
    _lib.check(lib.sp_score_matrix_f64(ctx.handle, _lib._ptr(m, C.c_double), m.shape[0], m.shape[1], m.shape[1],
                                       C.byref(out)))
    return np.float64(out.value)


def _score_big_sparse(ri, ci, v, shape):
    """A sparse matrix whose smaller side exceeds what the dense route keeps in LDS (1024 compact rows): the flattening of
    12 and more taxa in the reference's default FlatFormat.sparse.  Re-expressed as a pattern table - key = row * 4^b + col
    with 4^a >= rows, 4^b >= cols is a table of a + b "taxa" whose flattening for the split (first a | last b) IS the
    matrix - and scored by the batched sparse route: the list kernels for count-derived values (value = count / N, as
    every table from an alignment has) up to 65535 cells, its big-table form for more cells or arbitrary non-negative
    values.  Needs a + b <= 31 (more than 16: big-table form only); otherwise None (the caller's dense route then reports the limit)."""
    from .batch import score_encoded
    from .device import DeviceAlignment, infer_counts

    rows, cols = int(shape[0]), int(shape[1])
    if min(rows, cols) <= 1024 or len(v) == 0:
        return None
    a = max(1, (max(rows - 1, 1).bit_length() + 1) // 2)
    b = max(1, (max(cols - 1, 1).bit_length() + 1) // 2)
    if a + b > 31:                                   # (the packed key row * 4^b + col has to fit 62 bits)
        return None
    keys = ri.astype(np.uint64) * np.uint64(4 ** b) + ci.astype(np.uint64)
    order = np.argsort(keys, kind="stable")
    keys, vals = keys[order], v[order]
    if len(keys) > 1 and (np.diff(keys.astype(np.int64)) == 0).any():
        return None                                  # duplicate cells: not a flattening
    inf = infer_counts(vals)
    if inf is not None:
        counts, n_sites = inf
        dev = DeviceAlignment.from_arrays(keys, None, a + b, counts=counts, n_sites=n_sites)
    else:                                            # arbitrary real values: the big-table form takes float weights
        if (vals < 0).any():
            return None                              # (its start block and trace assume non-negative weights only as a
                                                     #  heuristic, but negative cells are not a flattening: dense route)
        dev = DeviceAlignment.from_arrays(keys, vals, a + b, exact=False)
    taxa_arr = np.arange(a + b, dtype=np.int32)[None, :]
    try:
        scores, status = score_encoded(dev, taxa_arr, np.array([a], dtype=np.int32), _lib.SP_METHOD_FLATTENING)
    except NotImplementedError:
        return None
    return float(scores[0])


def flattening_rank_1_approximation(flattening, return_vectors=False, dont_compute_matrix=False):
    """Product of the marginals of a flattening (reference splitp/phylogenetics.py:332-341): r = column sums, c = row
    sums, approximation = r^T c.  Host bookkeeping on a matrix the caller already holds; same return forms."""
    m = np.array(flattening, dtype=np.float64) if not is_sparse(flattening) else flattening
    r = np.array([np.asarray(m.sum(axis=0)).ravel()])
    c = np.array([np.asarray(m.sum(axis=1)).ravel()])
    approximation = None if dont_compute_matrix else r.T @ c
    if return_vectors:
        return approximation, r.tolist()[0], c.tolist()[0]
    return approximation


def flattening_rank_k_approximation(split, alignment):
    """reference splitp/phylogenetics.py:343-361: for each letter x the column sums of the flattening with x banned on
    the row side and the row sums with x banned on the column side (constructions.py:94-99), summed outer products ->
    scipy.sparse matrix of shape (4^|B|, 4^|A|).  Taxa = sorted union of the halves (:344).

    The per-pattern row / column indices come from the device (sp_flatten_indices, once for all eight banned
    flattenings); banning is a digit count on those indices, the marginals are grouped sums."""
    from scipy.sparse import csr_matrix

    from .constructions import _digit_count, _indices
    from .device import as_device_alignment, normalise_split

    split = normalise_split(split)
    taxa = sorted(set(split[0]) | set(split[1]))
    al = as_device_alignment(alignment)
    where = {t: i for i, t in enumerate(taxa)}
    oa = np.array([where[s] for s in split[0]], dtype=np.int32)
    ob = np.array([where[s] for s in split[1]], dtype=np.int32)
    rows, cols = _indices(al, oa, ob)
    _, vals, _ = al.fetch()
    n_r, n_c = 4 ** len(oa), 4 ** len(ob)
    total = None
    for letter in "ACGT":
        keep_r = _digit_count(rows, len(oa), letter) <= 1          # row pattern holds the letter at most once
        keep_c = _digit_count(cols, len(ob), letter) <= 1
        col_sums = np.bincount(cols[keep_r], weights=vals[keep_r], minlength=n_c)     # sum(F banned on rows) over rows
        row_sums = np.bincount(rows[keep_c], weights=vals[keep_c], minlength=n_r)     # sum(F^T banned on cols)
        term = csr_matrix(col_sums[:, None]) @ csr_matrix(row_sums[None, :])
        total = term if total is None else total + term
    return total


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
