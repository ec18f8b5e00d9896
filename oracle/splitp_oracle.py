"""CPU oracle for the SplitP flattening / subflattening / split_score path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
and bench.py's `cpu_baseline` leg may import it.  The product path (splitp_amd/) never
imports anything from oracle/ and fails loudly when the HIP library is missing.

It is a restatement, written from the behaviour documented in SURVEY.md section 8a, of the
reference algorithm (js51/SplitP v0.3.2).  Each function cites the reference file:line it
follows.  Parity status: PINNED - checked in tests/test_oracle_golden.py against
  * the reference's own test vectors (tests/test_constructions.py:5-107, restated as
    arrays in tests/golden/ref4.npz), and
  * outputs of the real reference run in the build container by tools/make_goldens.py
    (tests/golden/n10_L10k.npz, n10_L100k.npz, n16_L4k.npz, degenerate.npz).

Two layers:
  1. "loops" functions - the reference's algorithm restated step for step (per-pattern
     Python loops, dict-of-dicts, scipy SVD).  These are what bench.py times as the CPU
     baseline ("port"), because their cost structure is the reference's.
  2. "packed" functions - the same maths vectorised over (key, value) arrays with NumPy,
     used as the checker at sizes where the loops would take minutes.  tests check
     layer 2 == layer 1 on the goldens.
"""
from __future__ import annotations

from math import sqrt

import numpy as np
import scipy.linalg
import scipy.sparse
import scipy.sparse.linalg

# reference: splitp/constants.py:7-8 (state order A,C,G,T -> digits 0..3)
STATES = ("A", "C", "G", "T")
DIGIT = {s: i for i, s in enumerate(STATES)}

# reference: splitp/constructions.py:143-147 - the "banned" pairs give sign +1, all other
# (label char, table char) pairs give -1.  Written out as the 4x4 sign table
# SIGN[label][table] (rows/cols in A,C,G,T order); it is a Hadamard matrix.
SIGN = np.array(
    [
        [+1, -1, -1, +1],  # label A
        [+1, +1, -1, -1],  # label C
        [+1, -1, +1, -1],  # label G
        [+1, +1, +1, +1],  # label T
    ],
    dtype=np.int64,
)


# --------------------------------------------------------------------------------------
# layer 1: faithful restatement (loops)
# --------------------------------------------------------------------------------------
def index_of(chars) -> int:
    """Base-4 value of a pattern substring, first character most significant.

    reference: splitp/constructions.py:166-171 (reverses the string and sums 4**o * digit).
    Characters outside ACGT raise KeyError, as in the reference (:170).

    The same operations as the reference on purpose (a power per digit, not Horner's rule): this layer is also the CPU
    baseline bench.py times, and a cheaper restatement would flatter the reference (47 % of its flattening time is
    spent here; tests/test_oracle_golden.py::test_oracle_wall_time_tracks_the_reference).
    """
    text = chars if isinstance(chars, str) else "".join(chars)
    value = 0
    for power, ch in enumerate(reversed(text)):
        value += (4 ** power) * DIGIT[ch]
    return value


def _normalise_split(split):
    # reference: constructions.py:19-20 - "01|23" -> ["01", "23"]
    if isinstance(split, str):
        split = split.split("|")
    return split


def _taxa_of(split, table):
    # reference: constructions.py:21-24 - .taxa attribute, else sorted union of the halves
    try:
        return table.taxa
    except AttributeError:
        return sorted(set.union(*map(set, split)))


def flattening(split, table, fmt="sparse"):
    """reference: splitp/constructions.py:7-28.  fmt is 'sparse' or 'reduced'
    (the names of the reference's FlatFormat members, enums.py:23-27); any other
    format returns None, like the reference."""
    split = _normalise_split(split)
    taxa = _taxa_of(split, table)
    fmt = getattr(fmt, "name", fmt)
    if fmt == "sparse":
        return sparse_flattening_loops(split, table, taxa)
    if fmt == "reduced":
        return reduced_flattening_loops(split, table, taxa)
    return None


def reduced_flattening_loops(split, table, taxa):
    """reference: splitp/constructions.py:31-55.

    Rows/cols that occur, ascending key order; values ASSIGNED (last pattern wins on a
    collision, which only happens when the split does not cover every taxon)."""
    split = _normalise_split(split)
    where = {t: i for i, t in enumerate(taxa)}
    left = [where[t] for t in split[0]]
    right = [where[t] for t in split[1]]
    cells = {}
    seen_cols = set()
    for pattern, value in table.items():
        r = index_of("".join([str(pattern[i]) for i in left]))
        c = index_of("".join([str(pattern[i]) for i in right]))
        seen_cols.add(c)
        cells.setdefault(r, {})[c] = value
    col_rank = {c: j for j, c in enumerate(sorted(seen_cols))}
    out = np.zeros((len(cells), len(seen_cols)))
    for i, r in enumerate(sorted(cells)):
        for c, value in cells[r].items():
            out[i, col_rank[c]] = value
    return out


def sparse_flattening_loops(split, table, taxa, ban_row_patterns=None, ban_col_patterns=None):
    """reference: splitp/constructions.py:58-102 (the 'dok' branch, :86-102).

    Full logical shape (4^a, 4^b) float64 dok; an entry is zeroed when its row (col)
    substring contains the banned letter more than once (:94-99)."""
    split = _normalise_split(split)
    where = {t: i for i, t in enumerate(taxa)}
    left = [where[t] for t in split[0]]
    right = [where[t] for t in split[1]]
    out = scipy.sparse.dok_matrix((4 ** len(left), 4 ** len(right)))
    for pattern, value in table.items():
        rs = "".join(str(pattern[i]) for i in left)
        cs = "".join(str(pattern[i]) for i in right)
        banned = (ban_col_patterns is not None and cs.count(ban_col_patterns) > 1) or (
            ban_row_patterns is not None and rs.count(ban_row_patterns) > 1
        )
        out[index_of(rs), index_of(cs)] = 0 if banned else value
    return out


def subflattening_labels(length):
    """reference: splitp/constructions.py:174-189.  Position-major, then A,C,G at that
    position with T elsewhere; the all-T label last."""
    labels = []
    for pos in range(length):
        for ch in STATES[:-1]:
            labels.append("T" * pos + ch + "T" * (length - pos - 1))
    labels.append("T" * length)
    return labels


def subflattening_loops(split, table, data=None):
    """reference: splitp/constructions.py:108-163 (signed sums in table iteration order).

    `data` is accepted and populated with 'coeffs'/'labels' keys like the reference's
    cache (:120-127); the sign products are recomputed here (same values)."""
    try:
        taxa = table.taxa
    except AttributeError:
        # NOTE the reference takes the union BEFORE splitting a string split (:114-117), so a
        # string split on a plain dict counts '|' as a taxon and later raises KeyError (:198).
        taxa = sorted(set.union(*map(set, split)))
    where = {t: i for i, t in enumerate(taxa)}
    if data is not None:
        data.setdefault("coeffs", {})
        data.setdefault("labels", {})
    split = _normalise_split(split)
    a, b = len(split[0]), len(split[1])
    n = len(where)
    rows, cols = subflattening_labels(a), subflattening_labels(b)
    if data is not None:
        data["labels"].setdefault(a, rows)
        data["labels"].setdefault(b, cols)
    out = [[0] * (3 * b + 1) for _ in range(3 * a + 1)]
    for ri, rl in enumerate(rows):
        for ci, cl in enumerate(cols):
            # reference :192-198 - scatter the two labels into an n-char pattern by taxon index
            full = {}
            for half, lab in ((split[0], rl), (split[1], cl)):
                for k, taxon in enumerate(half):
                    full[where[taxon]] = lab[k]
            label = [full[i] for i in range(n)]  # KeyError if the split misses a taxon (:198)
            acc = 0
            for pattern, value in table.items():
                sign = 1
                for lc, tc in zip(label, pattern):
                    if SIGN[DIGIT[lc], DIGIT[tc]] < 0:
                        sign = -sign
                acc += sign * value
            out[ri][ci] = acc
    return np.array(out)


def is_sparse(matrix):
    """reference: splitp/matrix.py:4-5"""
    return scipy.sparse.issparse(matrix)


def frobenius_norm(matrix, data_table=None):
    """reference: splitp/matrix.py:7-14 (Python-level sums of squares)."""
    if data_table is not None:
        return sum(v**2 for _, v in data_table.itertuples(index=False)) ** 0.5
    if is_sparse(matrix):
        r, c = matrix.nonzero()
        return sum(matrix[i, j] ** 2 for i, j in zip(r, c)) ** 0.5
    return np.sqrt(sum(v**2 for v in np.nditer(matrix)))


def dense_split_score(matrix):
    """reference: splitp/phylogenetics.py:280-300.

    All min(R,C) singular values by LAPACK gesdd; (1 - sum s[:4]^2 / sum s^2) ** 0.5 with
    Python-level generator sums and NO clamp (nan when rounding drives it negative)."""
    s = list(scipy.linalg.svd(np.array(matrix), full_matrices=False, check_finite=False, compute_uv=False))
    m = min(matrix.shape)
    return (1 - sum(v**2 for v in s[0:4]) / sum(v**2 for v in s[0:m])) ** 0.5


def sparse_split_score(matrix):
    """reference: splitp/phylogenetics.py:303-312 (ARPACK top-4, Frobenius norm, clamp at 0)."""
    top = scipy.sparse.linalg.svds(matrix, 4, return_singular_vectors=False)
    norm = frobenius_norm(matrix)
    operand = 1 - sum(v**2 for v in top) / norm**2
    return sqrt(operand if operand > 0 else 0)


def split_score(matrix, return_singular_values=False, force_frob_norm_on_dense=False, data_table_for_frob_norm=None):
    """reference: splitp/phylogenetics.py:315-328.  The two boolean options are no-ops in
    the reference (misrouted positionally) and are ignored here too."""
    if is_sparse(matrix):
        return sparse_split_score(matrix)
    return dense_split_score(matrix)


def rank1_marginals(flattening):
    """reference: splitp/phylogenetics.py:332-341 with return_vectors=True, dont_compute_matrix=True:
    r = builtin sum over the rows (= column sums), c = builtin sum over the transposed rows (= row sums)."""
    r = np.array([sum(flattening)])
    c = np.array([sum(flattening.T)])
    return r.tolist()[0], c.tolist()[0]


def rank1_divergence(flattening):
    """reference: splitp/phylogenetics.py:364-373 (flattening_rank_1_approximation_divergence): Python loops over
    every cell, rows outer, columns inner, non-zero cells add f * log(f / (r[y] * c[x]))."""
    r, c = rank1_marginals(flattening)
    total = 0
    for x in range(len(c)):
        for y in range(len(r)):
            if flattening[x, y] != 0:
                total += flattening[x, y] * np.log(flattening[x, y] / (r[y] * c[x]))
    return total


def rank_k_approximation(split, table):
    """reference: splitp/phylogenetics.py:343-361 flattening_rank_k_approximation - for each letter the column sums of
    the flattening with that letter banned on the row side (builtin sum over the rows of the sparse matrix, :346-353) and
    the row sums with it banned on the column side (:354-360), then the sum over the letters of the outer products
    A^T B (:361): a (4^|B| x 4^|A|) sparse matrix.  Taxa = sorted union of the halves (:344)."""
    split = _normalise_split(split)
    taxa = sorted(set(split[0]) | set(split[1]))
    col_sums = [sum(sparse_flattening_loops(split, table, taxa, ban_row_patterns=ch)) for ch in "ACGT"]
    row_sums = [sum(sparse_flattening_loops(split, table, taxa, ban_col_patterns=ch).T) for ch in "ACGT"]
    return sum(a.T * b for a, b in zip(col_sums, row_sums))


def rank1_divergence_packed(keys, vals, n_taxa, order_a, order_b):
    """Vectorised form on a packed table (a cell of the flattening is one pattern): same value up to summation order."""
    rows, cols = flat_indices(keys, n_taxa, order_a, order_b)
    vals = np.asarray(vals, dtype=np.float64)
    _, ri = np.unique(rows, return_inverse=True)
    _, ci = np.unique(cols, return_inverse=True)
    rs = np.bincount(ri, weights=vals)
    cs = np.bincount(ci, weights=vals)
    nz = vals != 0
    return float(np.sum(vals[nz] * np.log(vals[nz] / (rs[ri[nz]] * cs[ci[nz]]))))


def all_splits(taxa, trivial=False, size=None):
    """Enumeration order of reference splitp/splits.py:27-59 (without randomise / string_format):
    sizes ascending from 2 (1 if trivial) to floor(n/2); combinations order inside a size;
    for the even split only combinations containing taxa[0]; taxa[0] always on the left."""
    from itertools import combinations

    taxa = list(taxa)
    n = len(taxa)
    sizes = [size] if size is not None else list(range(1 if trivial else 2, n // 2 + 1))
    for bal in sizes:
        even = bal == n / 2
        combos = combinations(taxa[1:], bal - 1) if even else combinations(taxa, bal)
        for left in combos:
            if even:
                left = (taxa[0],) + left
            right = tuple(sorted(set(taxa) - set(left), key=taxa.index))
            left = tuple(sorted(left, key=taxa.index))
            if taxa[0] in right:
                left, right = right, left
            yield (left, right)


# --------------------------------------------------------------------------------------
# layer 2: the same maths over packed arrays (vectorised NumPy) - checker for big cases
# --------------------------------------------------------------------------------------
def pack_table(table, n_taxa=None):
    """dict pattern -> value  =>  (keys uint64, values float64) in dict order.
    key = base-4 value of the whole pattern, taxon 0 most significant."""
    keys = np.empty(len(table), dtype=np.uint64)
    vals = np.empty(len(table), dtype=np.float64)
    for i, (p, v) in enumerate(table.items()):
        keys[i] = index_of(str(c) for c in p)
        vals[i] = v
    return keys, vals


def unpack_table(keys, vals, n_taxa):
    """inverse of pack_table (plain dict, same order)."""
    out = {}
    for k, v in zip(keys.tolist(), vals.tolist()):
        chars = []
        for t in range(n_taxa):
            chars.append(STATES[(k >> (2 * (n_taxa - 1 - t))) & 3])
        out["".join(chars)] = v
    return out


def digits_of(keys, n_taxa):
    """(D, n) uint8 array of base-4 digits, column t = taxon t."""
    keys = np.asarray(keys, dtype=np.uint64)
    shifts = (2 * (n_taxa - 1 - np.arange(n_taxa))).astype(np.uint64)
    return ((keys[:, None] >> shifts[None, :]) & np.uint64(3)).astype(np.uint8)


def flat_indices(keys, n_taxa, order_a, order_b):
    """row/col index of every pattern for the split (order_a | order_b), taxa given as
    indices in the order the split lists them (first listed = most significant digit).
    Same value as index_of() applied to the picked characters (constructions.py:39-40,:90-93)."""
    dg = digits_of(keys, n_taxa).astype(np.int64)
    rows = np.zeros(len(dg), dtype=np.int64)
    cols = np.zeros(len(dg), dtype=np.int64)
    for t in order_a:
        rows = rows * 4 + dg[:, t]
    for t in order_b:
        cols = cols * 4 + dg[:, t]
    return rows, cols


def reduced_flattening_packed(keys, vals, n_taxa, order_a, order_b):
    """constructions.py:31-55 vectorised (assumes a collision-free split, i.e. one that
    covers every taxon).  Returns (matrix, used_row_keys, used_col_keys)."""
    rows, cols = flat_indices(keys, n_taxa, order_a, order_b)
    ur, ri = np.unique(rows, return_inverse=True)
    uc, ci = np.unique(cols, return_inverse=True)
    out = np.zeros((len(ur), len(uc)), dtype=np.asarray(vals).dtype)
    out[ri, ci] = vals
    return out, ur, uc


def dense_flattening_packed(keys, vals, n_taxa, order_a, order_b):
    """Full 4^a x 4^b matrix (the .todense() of constructions.py:86-102)."""
    rows, cols = flat_indices(keys, n_taxa, order_a, order_b)
    out = np.zeros((4 ** len(order_a), 4 ** len(order_b)), dtype=np.asarray(vals).dtype)
    out[rows, cols] = vals
    return out


def sign_vectors(keys, n_taxa):
    """W in {+-1}^(D x (3n+1)): W[p, 3t+j] = SIGN[j, digit_t(p)] for j in A,C,G; last column 1
    (SURVEY A.3)."""
    dg = digits_of(keys, n_taxa)
    w = np.ones((len(dg), 3 * n_taxa + 1), dtype=np.int64)
    for t in range(n_taxa):
        for j in range(3):
            w[:, 3 * t + j] = SIGN[j, dg[:, t]]
    return w


def moment_matrix(keys, weights, n_taxa):
    """M = W^T diag(weights) W.  Exact integers when weights are integer counts."""
    w = sign_vectors(keys, n_taxa)
    weights = np.asarray(weights)
    if np.issubdtype(weights.dtype, np.integer):
        return (w * weights.astype(np.int64)[:, None]).T @ w
    return (w * weights[:, None]).T @ w


def subflattening_index(order, n_taxa):
    """Rows of M selected by one split half: [3t+j for t in order for j in 0..2] + [3n]."""
    idx = [3 * t + j for t in order for j in range(3)]
    idx.append(3 * n_taxa)
    return np.array(idx, dtype=np.int64)


def subflattening_packed(keys, weights, n_taxa, order_a, order_b):
    """constructions.py:108-163 through the second-moment identity."""
    m = moment_matrix(keys, weights, n_taxa)
    return m[np.ix_(subflattening_index(order_a, n_taxa), subflattening_index(order_b, n_taxa))]


def score_from_matrix_gram(matrix):
    """score via the Gram matrix over the smaller side: sqrt(max(0, 1 - top4(eig G)/trace G)).
    Mathematically equal to dense_split_score; used to check the HIP path's route."""
    m = np.asarray(matrix, dtype=np.float64)
    if m.shape[0] > m.shape[1]:
        m = m.T
    g = m @ m.T
    ev = np.linalg.eigvalsh(g)
    top = ev[::-1][:4].sum()
    tr = np.trace(g)
    if tr == 0:
        return float("nan")
    return sqrt(max(0.0, 1.0 - top / tr))


# --------------------------------------------------------------------------------------------------------------
# Simulator: exact distribution of the site patterns reference splitp/simulation.py:9-40 samples from,
# and a per-site restatement of that walk for small Monte-Carlo cross-checks.
def pattern_distribution(parent, leaf_taxon, trans, n_taxa):
    """P(pattern) for all 4^n patterns (index = base-4 value of the pattern, taxon 0 most significant).
    Tree in parents-first arrays: parent[i] < i, root = node 0 (uniform state, simulation.py:28); trans[i] is node i's
    4 x 4 matrix M with M[new, old] (a state is drawn from column `old`, simulation.py:17-18).  Pruning from the leaves up."""
    n_nodes = len(parent)
    pats = np.arange(4 ** n_taxa)
    digit = lambda t: (pats >> (2 * (n_taxa - 1 - t))) & 3          # noqa: E731
    like = [None] * n_nodes                                           # like[node][pattern, state of node]
    children = [[] for _ in range(n_nodes)]
    for i in range(1, n_nodes):
        children[parent[i]].append(i)
    for node in range(n_nodes - 1, -1, -1):
        if leaf_taxon[node] >= 0:
            lk = np.zeros((len(pats), 4))
            lk[np.arange(len(pats)), digit(leaf_taxon[node])] = 1.0
        else:
            lk = np.ones((len(pats), 4))
            for ch in children[node]:
                lk = lk * (like[ch] @ np.asarray(trans[ch], dtype=np.float64))   # sum_new like[new] M[new, old]
        like[node] = lk
    return like[0] @ np.full(4, 0.25)


def evolve_pattern_keys(parent, leaf_taxon, trans, n_taxa, n_sites, rng):
    """Per-site walk of simulation.py:9-40 (vectorised over sites, numpy generator): packed keys of n_sites patterns."""
    n_nodes = len(parent)
    state = np.zeros((n_nodes, n_sites), dtype=np.int64)
    state[0] = rng.integers(0, 4, size=n_sites)
    keys = np.zeros(n_sites, dtype=np.uint64)
    for node in range(1, n_nodes):
        m = np.asarray(trans[node], dtype=np.float64)
        cum = np.cumsum(m, axis=0)                                    # cumulative over the new state, per old state
        u = rng.random(n_sites) * cum[3, state[parent[node]]]
        old = state[parent[node]]
        state[node] = (u >= cum[0, old]).astype(np.int64) + (u >= cum[1, old]) + (u >= cum[2, old])
    for node in range(n_nodes):
        if leaf_taxon[node] >= 0:
            keys |= state[node].astype(np.uint64) << np.uint64(2 * (n_taxa - 1 - leaf_taxon[node]))
    return keys
