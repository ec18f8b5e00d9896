"""GPU tests of the direct finisher (csrc/finish.hip: Householder tridiagonalisation + Sturm multisection on the Gram
matrix - the always-answering counterpart of the reference's LAPACK gesdd, phylogenetics.py:280-300) and of the rule that
an iterative route's score is CERTIFIED OR FLAGGED, never silently wrong."""
import os

import numpy as np
import pytest
import scipy.sparse

from oracle import splitp_oracle as O
from tests.conftest import GOLDEN, mask_to_split, taxa_names

pytestmark = pytest.mark.gpu

SCORE_TOL = 1e-10
OPTIONS = (("force_big", 0), ("big_by_keys", 0), ("lds_cap", 0), ("wide_cap", 0), ("direct_finish", 1), ("direct_all", 0),
           ("direct_max_rows", 0), ("sort_three_launch", 0), ("eigen_block16", 0), ("sort_digit_bits", 0))


@pytest.fixture(scope="module")
def sp():
    import splitp_amd

    splitp_amd._lib.require_gpu()
    return splitp_amd


@pytest.fixture(autouse=True)
def _reset_context_options(sp):
    yield
    ctx = sp.get_context()
    for name, val in OPTIONS:
        ctx.set_option(name, val)


def _close(want, got):
    """score within 1e-10, or 1 - top4/trace within 8e-15 (its fp64 floor)"""
    return abs(want - got) <= SCORE_TOL or abs(want * want - got * got) <= 8e-15


def _regress_tables():
    d = np.load(os.path.join(GOLDEN, "regress_wide_flat_spectrum.npz"))
    for tag in ("a", "b"):
        n = int(d[tag + "_n"])
        names = taxa_names(n)
        splits = []
        for row in d[tag + "_left"]:
            left = [int(t) for t in row if t >= 0]
            splits.append((tuple(names[t] for t in left), tuple(names[t] for t in range(n) if t not in left)))
        yield tag, n, names, d[tag + "_keys"], d[tag + "_counts"], splits, d[tag + "_want"]


def test_regress_wide_block_flat_spectrum(sp):
    """The two tables of the second soak of round 3 (tools/gpu_fuzz_long.py seeds 71002 / 469 and 71004 / 665, regenerated on
    the CPU by tools/make_regress_wide_flat.py): random 3-letter tables of 14 / 12 taxa whose flattenings have a nearly flat
    spectrum behind the 4th value.  Round 3's 8-wide block returned two scores 1e-3 off with status "converged".
    Now, on every form of the chain: with the direct solver switched off a score is within tolerance OR flagged (status
    bit 0) - never an unflagged miss -, and with it (the default) every score is within tolerance and nothing is flagged."""
    ctx = sp.get_context()
    forms = (("default", {}), ("force_big", {"force_big": 1}), ("big_by_keys", {"force_big": 1, "big_by_keys": 1}),
             ("lds_cap_30k", {"lds_cap": 30000}), ("lds_cap_12k", {"lds_cap": 12000}))
    seen_flagged = 0
    for tag, n, names, keys, counts, splits, want in _regress_tables():
        dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=int(counts.sum()), taxa=names)
        for form, opts in forms:
            for name, val in OPTIONS:
                ctx.set_option(name, val)
            for name, val in opts.items():
                ctx.set_option(name, val)
            ctx.set_option("direct_finish", 0)
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                got, st = sp.score_splits(dev, splits, return_status=True)
            for i in range(len(splits)):
                assert (st[i] & 1) or _close(want[i], got[i]), (tag, form, i, want[i], got[i], hex(st[i]))
                assert not (st[i] & 2), (tag, form, i, hex(st[i]))
            seen_flagged += int(np.count_nonzero(st & 1))
            assert np.all(got[(st & 1) != 0] >= want[(st & 1) != 0] - 1e-9)     # flagged = upper estimate
            ctx.set_option("direct_finish", 1)
            got, st = sp.score_splits(dev, splits, return_status=True)
            assert not np.any(st & 3), (tag, form, [hex(v) for v in st])
            for i in range(len(splits)):
                assert _close(want[i], got[i]), (tag, form, i, want[i], got[i], hex(st[i]))
    assert seen_flagged > 0    # (the tables do exercise the flag: otherwise this test pins nothing)


def test_direct_solver_generic_matrices(sp):
    """split_score(matrix) with every matrix forced through the direct solver (option direct_all) against the oracle's
    LAPACK SVD: ragged shapes, both orientations, smaller sides of 5 ... 1500 rows (beyond the 1024 rows the block
    iteration's kernels hold - such matrices were SP_ELIMIT until round 4), flat and clustered spectra, scales."""
    ctx = sp.get_context()
    rng = np.random.default_rng(2024)
    ctx.set_option("direct_all", 1)
    for shape in [(5, 40), (40, 5), (6, 6), (16, 16), (17, 300), (63, 64), (65, 130), (200, 31), (257, 300), (700, 900)]:
        M = rng.integers(0, 7, size=shape).astype(np.float64)
        assert abs(sp.split_score(M) - O.dense_split_score(M)) <= SCORE_TOL, shape
    M = rng.standard_normal((120, 400))                                   # no gap anywhere
    assert abs(sp.split_score(M) - O.dense_split_score(M)) <= SCORE_TOL
    U, _ = np.linalg.qr(rng.standard_normal((90, 90)))
    V, _ = np.linalg.qr(rng.standard_normal((140, 90)))
    sv = np.array([5.0, 4.0, 3.0] + [2.0] * 6 + [2.0 - 1e-9] * 3 + list(np.linspace(1.9, 0.1, 78)))   # lambda_4 in a cluster
    M = (U * sv) @ V.T
    assert abs(sp.split_score(M) - O.dense_split_score(M)) <= SCORE_TOL
    M = np.where(rng.random((300, 500)) < 0.1, rng.integers(1, 1000, (300, 500)), 0).astype(np.float64)
    want = O.dense_split_score(M)
    for scale in (1e-8, 1e5, 1e-30, 1e30):
        assert abs(sp.split_score(M * scale) - want) <= SCORE_TOL, scale
        assert abs(sp.split_score(scipy.sparse.csr_matrix(M * scale)) - want) <= SCORE_TOL, scale
    assert sp.split_score(rng.standard_normal((3, 50))) == 0.0            # min(shape) <= 4
    assert np.isnan(sp.split_score(np.zeros((6, 9))))
    r = sp.split_score(np.outer(rng.standard_normal(30), rng.standard_normal(45)))     # rank 1
    assert r * r <= 1e-13
    ctx.set_option("direct_all", 0)
    # beyond the block iteration's 1024 rows: the direct solver takes over by itself
    M = rng.standard_normal((1100, 1300))                                 # (negative cells: not a flattening, no sparse route)
    assert abs(sp.split_score(M) - O.dense_split_score(M)) <= SCORE_TOL
    M = np.where(rng.random((1300, 1500)) < 0.02, rng.standard_normal((1300, 1500)), 0.0)
    want = O.dense_split_score(M)
    assert abs(sp.split_score(M) - want) <= SCORE_TOL
    assert abs(sp.split_score(scipy.sparse.csr_matrix(M)) - want) <= SCORE_TOL


@pytest.mark.parametrize("name", ["n10_L10k", "n10_L100k"])
def test_direct_solver_all_501_golden_scores(sp, golden, name):
    """The dense route with the direct solver in place of its block iteration: all 501 reference scores of both 10-taxon
    alignments (count table and float-weight table), and bit-identical when repeated."""
    g = golden(name)
    names = taxa_names(10)
    splits = [mask_to_split(int(m), 10, names) for m in g["masks"]]
    dev = sp.DeviceAlignment.from_table(O.unpack_table(g["keys"], g["probs"], 10), taxa=names)
    sp.get_context().set_option("direct_all", 1)
    got, st = sp.score_splits(dev, splits, route="dense", return_status=True)
    assert np.abs(got - g["scores"]).max() <= SCORE_TOL
    assert np.all((st & 7) == 4)                                          # status bit 2: scored by the direct solver
    again = sp.score_splits(dev, splits, route="dense")
    assert np.array_equal(got, again)
    probs = np.asarray(g["probs"], dtype=np.float64)
    dev_w = sp.DeviceAlignment.from_arrays(g["keys"], probs, 10, taxa=names, exact=False)
    got_w = sp.score_splits(dev_w, splits, route="dense")
    assert np.abs(got_w - g["scores"]).max() <= SCORE_TOL


def test_finish_flagged_abi_and_async(sp):
    """sp_finish_flagged (ABI 4) behind an asynchronous pass: a gapless 10-taxon table leaves flagged splits (status bit 0,
    upper estimates); the host step finishes exactly those (status bit 2, rows of the solved matrix in bits 8..), and the
    synchronous entry point does the same by itself.  A size cap (direct_max_rows) leaves what exceeds it flagged and
    reported (SP_ENOCONV / RuntimeWarning)."""
    import warnings

    import torch
    from splitp_amd import _lib, batch

    names = taxa_names(10)
    rng = np.random.default_rng(3)
    rk = np.unique(rng.integers(0, 4 ** 10, size=3000).astype(np.uint64))
    rc = rng.integers(1, 40, size=len(rk)).astype(np.int64)
    flat = sp.DeviceAlignment.from_arrays(rk, None, 10, counts=rc, n_sites=int(rc.sum()), taxa=names)
    taxa_arr, a_arr = sp.encode_all_splits(10)
    sub_t, sub_a = np.ascontiguousarray(taxa_arr[::25]), np.ascontiguousarray(a_arr[::25])
    splits = list(sp.all_splits(names))[::25]
    want = np.array([O.dense_split_score(O.reduced_flattening_packed(rk, rc.astype(np.float64), 10, [names.index(t) for t in s[0]],
                                                                     [names.index(t) for t in s[1]])[0]) for s in splits])
    sc = torch.zeros(len(sub_a), dtype=torch.float64, device="cuda")
    st = torch.zeros(len(sub_a), dtype=torch.int32, device="cuda")
    batch.score_encoded_async(flat, sub_t, sub_a, _lib.SP_METHOD_FLATTENING, sc.data_ptr(), st.data_ptr())
    torch.cuda.synchronize()
    s_h, t_h = sc.cpu().numpy(), st.cpu().numpy()
    flagged = (t_h & 1) != 0
    assert flagged.sum() > 0 and not np.any(t_h & 2)
    assert np.all(np.abs(s_h - want)[~flagged] <= SCORE_TOL)             # certified = right
    assert np.all(s_h[flagged] >= want[flagged] - 1e-9)                  # flagged = upper estimate
    keep = s_h.copy()
    assert batch.finish_async(flat, sub_t, sub_a, s_h, t_h) == int(flagged.sum())
    assert not np.any(t_h & 3) and np.all((t_h[flagged] & 4) != 0) and np.array_equal(s_h[~flagged], keep[~flagged])
    assert np.abs(s_h - want).max() <= SCORE_TOL
    got, st2 = sp.score_splits(flat, splits, return_status=True)          # the synchronous entry point: finished inside
    assert not np.any(st2 & 3) and np.abs(got - want).max() <= SCORE_TOL
    # the dense route behind the asynchronous entry: no status fetch, no synchronisation in the call - what its 4-wide eigen
    # kernel cannot certify on this table comes back flagged (bit 1: handed back) and is finished by the same host step
    scd = torch.zeros(len(sub_a), dtype=torch.float64, device="cuda")
    std = torch.zeros(len(sub_a), dtype=torch.int32, device="cuda")
    batch.score_encoded_async(flat, sub_t, sub_a, _lib.SP_METHOD_FLATTENING_DENSE, scd.data_ptr(), std.data_ptr())
    torch.cuda.synchronize()
    sd_h, td_h = scd.cpu().numpy(), std.cpu().numpy()
    flagged_d = (td_h & 3) != 0
    assert flagged_d.sum() > 0 and np.all(np.abs(sd_h - want)[~flagged_d] <= SCORE_TOL)
    assert batch.finish_async(flat, sub_t, sub_a, sd_h, td_h) == int(flagged_d.sum())
    assert not np.any(td_h & 3) and np.abs(sd_h - want).max() <= SCORE_TOL
    got_d, st_d = sp.score_splits(flat, splits, route="dense", return_status=True)   # synchronous: finished inside
    assert not np.any(st_d & 3) and np.abs(got_d - want).max() <= SCORE_TOL
    ctx = sp.get_context()
    ctx.set_option("direct_max_rows", 100)                                # the 4|6 and 5|5 splits exceed it
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        got3, st3 = sp.score_splits(flat, splits, return_status=True)
    small = np.array([min(len(s[0]), len(s[1])) <= 3 for s in splits])
    assert not np.any(st3[small] & 3) and np.abs(got3 - want)[small].max() <= SCORE_TOL
    if np.any(st3 & 1):
        assert any(issubclass(w.category, RuntimeWarning) for w in caught)
        assert np.all(got3[(st3 & 1) != 0] >= want[(st3 & 1) != 0] - 1e-9)


def test_direct_solver_bigger_sides_12_taxa(sp):
    """12-taxon random tables without structure (sides of up to 4096 ids: beyond what the dense route's block iteration
    ever took): with the chain's budget cut to a few half products (wide_cap) EVERY split ends in the direct solver -
    compare with the oracle's SVD of the reduced flattening."""
    from tests.test_gpu_parity import _copy_mutate_table

    rng = np.random.default_rng(5)
    n = 12
    keys, counts = _copy_mutate_table(rng, n, 20000, 3)
    names = taxa_names(n)
    dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=int(counts.sum()), taxa=names)
    splits = []
    for k in (2, 3, 4, 5, 6, 6, 5, 4):
        left = sorted(rng.choice(n, size=k, replace=False).tolist())
        splits.append((tuple(names[t] for t in left), tuple(names[t] for t in range(n) if t not in left)))
    want = np.array([O.dense_split_score(O.reduced_flattening_packed(keys, counts.astype(np.float64), n, [names.index(t) for t in s[0]],
                                                                     [names.index(t) for t in s[1]])[0]) for s in splits])
    got, st = sp.score_splits(dev, splits, return_status=True)
    assert not np.any(st & 3) and np.abs(got - want).max() <= SCORE_TOL
    sp.get_context().set_option("wide_cap", 6)
    got2, st2 = sp.score_splits(dev, splits, return_status=True)
    assert not np.any(st2 & 3) and np.abs(got2 - want).max() <= SCORE_TOL
    assert np.count_nonzero(st2 & 4) >= np.count_nonzero(st & 4)


def test_radix_sort_direct(sp):
    """csrc/radix_sort.h through its test entry: random 32- and 64-bit keys with values, one and many segments,
    segment lengths around the 4096-key tile (4095 / 4096 / 4097, a single key, 3 tiles + 1), key widths that are no multiple
    of the 8-bit digit (17, 33, 41 bits) - against numpy's stable argsort per segment; the one-sweep (decoupled look-back)
    form and round 3's three-launch form must agree bit for bit."""
    import ctypes as C

    from splitp_amd import _lib

    ctx = sp.get_context()
    lib = ctx._lib
    rng = np.random.default_rng(99)
    cases = [(4, 1, 1, 9), (4, 4095, 1, 17), (4, 4096, 3, 17), (4, 4097, 5, 24), (4, 12289, 7, 32), (8, 4097, 3, 33),
             (8, 100_000, 2, 41), (8, 5000, 64, 40), (4, 300_000, 1, 29), (4, 777, 257, 13)]
    for key_bytes, seg_len, n_seg, bits in cases:
        n = seg_len * n_seg
        keys = rng.integers(0, 1 << bits, size=n, dtype=np.uint64)
        if seg_len > 2000:
            keys[: n // 3] &= np.uint64(0xFF)          # heavy duplicates: stability matters
        vals = np.arange(n, dtype=np.uint32)
        outs = []
        for legacy, digit_bits in ((0, 8), (0, 9), (1, 0)):          # one-sweep with 8- / 9-bit digits, round 3's three-launch form
            ctx.set_option("sort_three_launch", legacy)
            ctx.set_option("sort_digit_bits", digit_bits)
            ko, vo = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint32)
            _lib.check(lib.sp_debug_radix_sort(ctx.handle, _lib._ptr(keys, C.c_uint64), _lib._ptr(vals, C.c_uint32), key_bytes,
                                               seg_len, n_seg, bits, _lib._ptr(ko, C.c_uint64), _lib._ptr(vo, C.c_uint32)))
            outs.append((ko, vo))
        ctx.set_option("sort_three_launch", 0)
        ctx.set_option("sort_digit_bits", 0)
        assert np.array_equal(outs[0][0], outs[2][0]) and np.array_equal(outs[0][1], outs[2][1])
        want_v = np.concatenate([s * seg_len + np.argsort(keys[s * seg_len:(s + 1) * seg_len], kind="stable")
                                 for s in range(n_seg)]).astype(np.uint32)
        assert np.array_equal(outs[0][1], want_v), (key_bytes, seg_len, n_seg, bits)
        assert np.array_equal(outs[0][0], keys[want_v]), (key_bytes, seg_len, n_seg, bits)
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
        ko = np.zeros(n, dtype=np.uint64)                          # keys only
        _lib.check(lib.sp_debug_radix_sort(ctx.handle, _lib._ptr(keys, C.c_uint64), None, key_bytes, seg_len, n_seg, bits,
                                           _lib._ptr(ko, C.c_uint64), None))
        assert np.array_equal(ko, keys[want_v])


def test_node_api_one_device(sp, golden):
    """sp_node_* (csrc/node.hip: one process drives the node's GPUs, RCCL loaded on first use): on the one-GPU box the node has
    one device - the whole path runs (ncclCommInitAll, shard = everything, ncclAllGather with one rank, un-permute) and must
    return the single-GPU call's results bit for bit, for the flattening and the subflattening route; asking for more
    devices than are visible is answered with an error, not a crash."""
    import ctypes as C

    from splitp_amd import _lib

    g = golden("n10_L100k")
    keys = np.asarray(g["keys"], dtype=np.uint64)
    probs = np.asarray(g["probs"], dtype=np.float64)
    counts = np.rint(probs * 100_000).astype(np.int64)
    node = sp.NodeScorer(1)
    assert node.n_devices == 1
    try:
        got, st = node.score_all_splits(keys, 10, counts=counts, n_sites=100_000, return_status=True)
        assert got.shape == (501,) and not np.any(st & 3)
        assert np.abs(got - g["scores"]).max() <= SCORE_TOL
        dev = sp.DeviceAlignment.from_arrays(keys, None, 10, counts=counts, n_sites=100_000, taxa=taxa_names(10))
        assert np.array_equal(got, sp.score_all_splits(dev))
        sub = node.score_all_splits(keys, 10, counts=counts, n_sites=100_000, method=sp.Method.subflattening)
        assert np.array_equal(sub, sp.score_all_splits(dev, method=sp.Method.subflattening))
        triv = node.score_all_splits(keys, 10, counts=counts, n_sites=100_000, method=sp.Method.subflattening, trivial=True, size=None)
        assert np.array_equal(triv, sp.score_all_splits(dev, method=sp.Method.subflattening, trivial=True))
        w = node.score_all_splits(keys, 10, weights=probs, method=sp.Method.subflattening)        # float-weight table
        assert np.abs(w - sub).max() <= 1e-12
    finally:
        node.close()
    # P = 2, 3, 8 ranks emulated on this one device (test mode: shards, packing and un-permuting of a real multi-rank run)
    for ranks in (2, 3, 8):
        em = sp.NodeScorer(-ranks)
        assert em.n_devices == ranks
        try:
            assert np.array_equal(em.score_all_splits(keys, 10, counts=counts, n_sites=100_000), got)
            assert np.array_equal(em.score_all_splits(keys, 10, counts=counts, n_sites=100_000, method=sp.Method.subflattening), sub)
            assert np.array_equal(em.score_all_splits(keys, 10, counts=counts, n_sites=100_000, method=sp.Method.subflattening,
                                                      size=5), sp.score_all_splits(dev, method=sp.Method.subflattening, size=5))
        finally:
            em.close()
    h = C.c_void_p()
    lib = _lib.load()
    assert lib.sp_node_create(_lib.device_count() + 1, C.byref(h)) == _lib.SP_EINVAL
    assert b"visible" in lib.sp_last_error()


def test_float_weight_table_12_taxa_ends_in_direct_solver(sp):
    """A 12-taxon table with real-valued weights (not count / N: the big-table form's float path) and no gap behind the 4th
    singular value: the 4-wide and 8-wide blocks certify next to nothing, the flagged splits are finished by the direct
    solver on the fp64 Gram matrix of the WEIGHTS - every score within 1e-10 of the oracle, nothing flagged, no warning."""
    import warnings

    from tests.test_gpu_parity import _copy_mutate_table

    rng = np.random.default_rng(12)
    n = 12
    keys, counts = _copy_mutate_table(rng, n, 20000, 3)
    names = taxa_names(n)
    w = counts / float(counts.sum()) * (1.0 + 1e-3 * rng.random(len(counts)))
    dev = sp.DeviceAlignment.from_arrays(keys, w, n, taxa=names, exact=False)
    splits = []
    for k in (2, 3, 4, 5, 6, 6, 5, 4, 3, 6):
        left = sorted(rng.choice(n, size=k, replace=False).tolist())
        splits.append((tuple(names[t] for t in left), tuple(names[t] for t in range(n) if t not in left)))
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        got, st = sp.score_splits(dev, splits, return_status=True)
    assert not any(issubclass(c.category, RuntimeWarning) for c in caught)
    want = np.array([O.dense_split_score(O.reduced_flattening_packed(keys, w, n, [names.index(t) for t in s[0]],
                                                                     [names.index(t) for t in s[1]])[0]) for s in splits])
    assert not np.any(st & 3) and np.count_nonzero(st & 4) >= 5
    assert np.abs(got - want).max() <= SCORE_TOL
