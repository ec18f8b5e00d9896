"""flattening / subflattening - drop-in for reference splitp/constructions.py, on MI355X.

Same names, argument order, defaults and return types as the reference:
    flattening(split, pattern_probabilities, flattening_format=FlatFormat.sparse)   constructions.py:7
    subflattening(split, pattern_probabilities, data=None)                          constructions.py:108
    sparse_flattening_with_banned_patterns(split, table, taxa, ban_row, ban_col)    constructions.py:105
The per-pattern Python loops run as HIP kernels (splitp_amd/csrc/flatten.hip, subflat.hip)
behind the C ABI of include/splitp_hip.h; results come back as the same host objects the
reference returns (scipy dok_matrix / numpy ndarray, float64)."""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np
from scipy.sparse import coo_matrix, dok_matrix

from . import _lib
from .constants import DNA_state_space_dict
from .device import as_device_alignment, normalise_split, resolve_split
from .enums import FlatFormat


def _orders(split, table, al):
    split = normalise_split(split)
    oa, ob = resolve_split(split, table, al.n_taxa)
    return oa, ob


def _indices(al, oa, ob):
    d = len(al)
    rows = np.empty(d, dtype=np.int64)
    cols = np.empty(d, dtype=np.int64)
    _lib.check(al.ctx._lib.sp_flatten_indices(
        al.handle, _lib._ptr(oa, C.c_int32), len(oa), _lib._ptr(ob, C.c_int32), len(ob),
        _lib._ptr(rows, C.c_int64), _lib._ptr(cols, C.c_int64)))
    return rows, cols


def flattening(split, pattern_probabilities, flattening_format=FlatFormat.sparse):
    """Compute the flattening of a split given a pattern probability dictionary.

    reference: splitp/constructions.py:7-28.  `pattern_probabilities` may be the reference's
    dict / Alignment or a splitp_amd.DeviceAlignment (already resident in HBM).  Any format other
    than FlatFormat.sparse / FlatFormat.reduced returns None, like the reference."""
    if flattening_format is not FlatFormat.sparse and flattening_format is not FlatFormat.reduced:
        name = getattr(flattening_format, "name", None)
        if name not in ("sparse", "reduced"):  # accept the reference's own enum members by name
            return None
        flattening_format = FlatFormat[name]
    al = as_device_alignment(pattern_probabilities)
    oa, ob = _orders(split, pattern_probabilities, al)
    if flattening_format is FlatFormat.sparse:
        return _sparse(al, oa, ob)
    return _reduced(al, oa, ob)


class Flattening(np.ndarray):
    """The reduced flattening as the reference returns it - a fresh, WRITABLE float64 ndarray (constructions.py:51-55) -
    that also remembers where it came from: (device-resident alignment, split).  `split_score(F)` then scores the split on
    the device from the resident table instead of uploading F again ("API impedance"): the unchanged
    README loop `split_score(flattening(split, table, FlatFormat.reduced))` keeps its semantics and loses the second PCIe
    trip.

    The remembered origin cannot go stale: it carries a checksum of the contents (128-bit xxh3 of the bytes, ~0.25 ms for a
    690 x 690 matrix), recomputed when the origin is asked for - an array edited in
    place (`F /= F.sum()`, `F[i, j] = 0`) no longer matches and is scored as the generic matrix it now is.  Anything
    derived from it - slices, arithmetic, copies, pickles - is a plain result with no origin."""

    _sp_origin = None
    _sp_check = None
    _sp_pending = None

    def __array_finalize__(self, obj):
        self._sp_origin = None
        self._sp_check = None
        self._sp_pending = None

    def __reduce__(self):      # pickling / copy.deepcopy: a plain array
        return np.asarray(self).__reduce__()


try:                                    # xxh3: ~16 GB/s on one core, no thread pool involved
    import xxhash as _xxhash
except Exception:                       # pragma: no cover - the image has it; hashlib is the portable stand-in
    _xxhash = None
import hashlib as _hashlib


def _content_check(arr):
    """Checksum of a C-contiguous float64 array's bytes (128-bit xxh3; blake2b where xxhash is missing) + its size.
    (Round 3 used a dot product with fixed pseudo-random weights: numpy hands that to the BLAS thread pool, and on a
    many-core host 476 k elements took 1.5 - 20 ms a call - 70 % of the drop-in loop, `tools/gpu_dropin_profile.py`.)"""
    flat = np.ascontiguousarray(arr).reshape(-1)
    if _xxhash is not None:
        return _xxhash.xxh3_128_intdigest(flat), flat.size
    return _hashlib.blake2b(memoryview(flat), digest_size=16).digest(), flat.size


def flattening_origin(matrix):
    """(DeviceAlignment, order_a, order_b) if `matrix` is an untouched result of flattening(..., FlatFormat.reduced)."""
    if type(matrix) is Flattening and matrix._sp_origin is not None and matrix.flags.c_contiguous:
        if _content_check(matrix) == matrix._sp_check:
            return matrix._sp_origin
        matrix._sp_origin = None        # edited in place: from now on an ordinary matrix
        matrix._sp_pending = None
    return None


# ---- the README loop's score, prefetched (tools/gpu_dropin_profile.py, round 4: 0.74 -> 0.55 ms a split) ----
# `split_score(flattening(split, table, FlatFormat.reduced))` asks for the score of a split the library has just been told
# about: once a caller has shown that pattern (split_score consumed an untouched Flattening of this table), the next
# flattening() enqueues the score of ITS split behind the fetch - sp_score_splits_async, the same kernels as the
# synchronous call - and the GPU works on it while the host checksums the matrix.  split_score(F) then only waits for the
# context's stream.  The score still comes from the resident table and still only when F's contents are untouched
# (flattening_origin); a caller who stops scoring costs two unused launches before the credit runs out.
# (Measured beside it and dropped: the matrix fetched into pinned host memory from torch's caching host allocator -
# +14 % on one box, -5 % on the next, `profiles/r04_dropin_profile.txt`; pageable numpy memory stays.)
PREFETCH_SCORES = True                 # module switch (tests compare both ways)
_PREFETCH_SLOTS = 64
_prefetchers = {}


class _ScorePrefetch:
    """Per-device ring of (score, status) slots: 16 bytes in HBM the asynchronous entry writes, and a pinned word pair the
    slot is read through.  Ordering is the context's own stream (a context created under torch's default stream runs on
    a private non-blocking stream - nothing torch enqueues is ordered with it): collect() synchronises THAT stream."""

    def __init__(self, device):
        import torch

        self.dev = torch.zeros((_PREFETCH_SLOTS, 2), dtype=torch.float64, device=torch.device("cuda", device))
        self.host = torch.zeros(2, dtype=torch.float64).pin_memory()
        self.host_np = self.host.numpy()
        self.ticket = [-1] * _PREFETCH_SLOTS
        self.issued = 0
        self._lock = threading.Lock()

    def issue(self, al, oa, ob):
        from .batch import score_encoded_async

        with self._lock:        # (ctypes calls release the GIL: two threads must never draw the same ticket / slot)
            t = self.issued
            self.issued = t + 1
        slot = t % _PREFETCH_SLOTS
        taxa = np.ascontiguousarray(np.concatenate([oa, ob])[None, :], dtype=np.int32)
        a = np.array([len(oa)], dtype=np.int32)
        base = self.dev.data_ptr() + slot * 16
        self.ticket[slot] = -1
        score_encoded_async(al, taxa, a, _lib.SP_METHOD_FLATTENING, base, base + 8)
        self.ticket[slot] = t
        return (self, t, taxa, a)

    def collect(self, al, pending):
        """np.float64 score, or None when the slot has been reused since (the caller scores synchronously)."""
        from .batch import finish_async, warn_unconverged

        _, t, taxa, a = pending
        slot = t % _PREFETCH_SLOTS
        if self.ticket[slot] != t:
            return None
        al.ctx.synchronize()                        # the kernels behind the fetch are done (usually long ago)
        self.host.copy_(self.dev[slot])             # 16 bytes, blocking
        sc = self.host_np[0:1].copy()
        st = self.host_np[1:2].view(np.int32)[0:1].copy()
        if st[0] & 3:                       # no certificate on the device: the library's direct solver answers (batch.finish_async)
            finish_async(al, taxa, a, sc, st)
        warn_unconverged(st, stacklevel=4)
        return np.float64(sc[0])


def _prefetch_score(al, oa, ob):
    if not PREFETCH_SCORES or getattr(al, "_sp_prefetch_credit", 0) <= 0:
        return None
    al._sp_prefetch_credit -= 1
    try:
        pf = _prefetchers.get(al.ctx.device)
        if pf is None:
            pf = _prefetchers[al.ctx.device] = _ScorePrefetch(al.ctx.device)
        return pf.issue(al, oa, ob)
    except Exception:       # a prefetch that cannot be enqueued is no error: split_score(F) makes the call itself and reports
        al._sp_prefetch_credit = 0
        return None


def take_prefetched_score(matrix, al):
    """The score flattening() enqueued for this untouched Flattening, if any (called by split_score after the origin
    check).  Also records that this table's flattenings are being scored: the next one is prefetched."""
    al._sp_prefetch_credit = 2
    pending = matrix._sp_pending
    matrix._sp_pending = None
    if pending is None:
        return None
    return pending[0].collect(al, pending)


def _reduced(al, oa, ob):
    # reference: constructions.py:31-55
    r, c = C.c_int64(), C.c_int64()
    lib = al.ctx._lib
    _lib.check(lib.sp_flatten_reduced_prepare(al.handle, _lib._ptr(oa, C.c_int32), len(oa), _lib._ptr(ob, C.c_int32),
                                              len(ob), C.byref(r), C.byref(c)))
    # (the fetch overwrites every cell of a non-empty matrix; an empty one has no cells)
    out = np.empty((r.value, c.value), dtype=np.float64)
    _lib.check(lib.sp_flatten_reduced_fetch(al.handle, _lib._ptr(out, C.c_double), None, None))
    oa32, ob32 = np.array(oa, dtype=np.int32), np.array(ob, dtype=np.int32)
    pending = _prefetch_score(al, oa32, ob32)      # the GPU scores the split while the host checksums the matrix
    out = out.view(Flattening)
    out._sp_origin = (al, oa32, ob32)
    out._sp_pending = pending
    out._sp_check = _content_check(out)
    return out


def _sparse(al, oa, ob, ban_row_patterns=None, ban_col_patterns=None):
    # reference: constructions.py:58-102 (dok branch)
    rows, cols = _indices(al, oa, ob)
    _, vals, _ = al.fetch()
    shape = (4 ** len(oa), 4 ** len(ob))
    if ban_row_patterns is not None or ban_col_patterns is not None:
        vals = vals.copy()
        if ban_row_patterns is not None:
            vals[_digit_count(rows, len(oa), ban_row_patterns) > 1] = 0.0
        if ban_col_patterns is not None:
            vals[_digit_count(cols, len(ob), ban_col_patterns) > 1] = 0.0
    keep = vals != 0  # a dok assignment of 0 stores nothing
    out = coo_matrix((vals[keep], (rows[keep], cols[keep])), shape=shape, dtype=np.float64).todok()
    if not isinstance(out, dok_matrix):
        out = dok_matrix(out)
    return out


def _digit_count(index, length, letter):
    """How many of the `length` base-4 digits of each index equal the digit of `letter`
    (the reference's row_pattern.count(letter), constructions.py:94-99)."""
    if len(str(letter)) != 1:
        # str.count of a multi-character pattern: fall back to explicit strings (rare, host-side option)
        out = np.zeros(len(index), dtype=np.int64)
        for i, v in enumerate(index.tolist()):
            s = "".join("ACGT"[(v >> (2 * (length - 1 - t))) & 3] for t in range(length))
            out[i] = s.count(letter)
        return out
    d = DNA_state_space_dict[letter]
    cnt = np.zeros(len(index), dtype=np.int64)
    for t in range(length):
        cnt += ((index >> (2 * t)) & 3) == d
    return cnt


def sparse_flattening_with_banned_patterns(split, pattern_probabilities, taxa, ban_row_patterns=None,
                                           ban_col_patterns=None):
    """reference: constructions.py:58-105 (alias of __sparse_flattening with an explicit taxa list)."""
    al = as_device_alignment(pattern_probabilities)
    split = normalise_split(split)
    where = {t: i for i, t in enumerate(taxa)}
    oa = np.array([where[s] for s in split[0]], dtype=np.int32)
    ob = np.array([where[s] for s in split[1]], dtype=np.int32)
    return _sparse(al, oa, ob, ban_row_patterns, ban_col_patterns)


def subflattening_labels(length):
    """Row / column labels of a subflattening (reference constructions.py:174-189)."""
    out = []
    for i in range(length):
        for ch in "ACG":
            out.append("T" * i + ch + "T" * (length - i - 1))
    out.append("T" * length)
    return out


def subflattening(split, pattern_probabilities, data=None):
    """Signed-sum subflattening, (3|A|+1) x (3|B|+1) float64 ndarray.

    reference: splitp/constructions.py:108-163.  Computed as a sub-block of the alignment's
    signed second-moment matrix (one device contraction per alignment, cached on the device
    table).  `data`, if given, receives the reference's cache keys ("coeffs", "labels"); the
    coefficient cache itself is not needed here and stays empty.

    Deviation (documented): for a string split on a plain dict the reference counts '|' as a taxon
    (constructions.py:114-117) and then raises KeyError; here the split is normalised first."""
    al = as_device_alignment(pattern_probabilities)
    split = normalise_split(split)
    oa, ob = resolve_split(split, pattern_probabilities, al.n_taxa)
    if data is not None:
        data.setdefault("coeffs", {})
        labels = data.setdefault("labels", {})
        labels.setdefault(len(oa), subflattening_labels(len(oa)))
        labels.setdefault(len(ob), subflattening_labels(len(ob)))
    if len(oa) + len(ob) != al.n_taxa:
        raise KeyError(len(oa) + len(ob))  # the reference's __reconstruct_pattern raises KeyError here (:198)
    out = np.empty((3 * len(oa) + 1, 3 * len(ob) + 1), dtype=np.float64)
    _lib.check(al.ctx._lib.sp_subflatten(al.handle, _lib._ptr(oa, C.c_int32), len(oa), _lib._ptr(ob, C.c_int32),
                                         len(ob), _lib._ptr(out, C.c_double)))
    return out
