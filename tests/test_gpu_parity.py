"""GPU parity tests: the HIP path (through the C ABI) against the oracle and the golden vectors.

Bars: flattening values / indices / integer counts bit-exact; subflattening
integer moments exact and value / N within 1e-13 of the matrix scale; fp64 scores within 1e-10
absolute (all golden scores are >= 1e-3), score^2 within 1e-13 for degenerate inputs."""
import numpy as np
import pytest
import scipy.sparse

from oracle import splitp_oracle as O
from tests.conftest import mask_to_split, taxa_names

pytestmark = pytest.mark.gpu

SCORE_TOL = 1e-10

REF4_SPLITS = [({0, 1}, {2, 3}), ({0, 2}, {1, 3}), ({0, 3}, {1, 2})]
REF4_TABLE = {"ATCG": 2 / 5, "GATC": 1 / 5, "CGAT": 1 / 5, "TCGA": 1 / 5}


@pytest.fixture(scope="module")
def sp():
    import splitp_amd

    splitp_amd._lib.require_gpu()
    return splitp_amd


@pytest.fixture(autouse=True)
def _reset_context_options(sp):
    """Test switches (sp_ctx_set_option) never leak from one test into the next."""
    yield
    ctx = sp.get_context()
    for name, val in (("force_big", 0), ("big_by_keys", 0), ("subscore_jacobi", 0), ("divergence_global", 0),
                      ("hist_sort", -1), ("lds_cap", 0), ("wide_cap", 0), ("direct_finish", 1), ("direct_all", 0),
                      ("direct_max_rows", 0), ("subscore_pair", 1), ("subscore_waves", 0), ("moments_valu", 0)):
        ctx.set_option(name, val)


def test_native_library_is_loaded(sp):
    # the product path is the in-tree .so, nothing else
    import os

    assert os.path.exists(sp._lib.LIB_PATH)
    assert sp._lib.device_count() >= 1
    with open("/proc/self/maps") as f:
        assert "libsplitp_hip.so" in f.read()


# ---------------------------------------------------------------- the reference's own tests, restated
def test_flattening_reference_vectors(sp):
    F = np.zeros((16, 16))
    F[3, 6] = .4; F[6, 3] = .2; F[8, 13] = .2; F[13, 8] = .2
    got = sp.flattening(REF4_SPLITS[0], REF4_TABLE)
    assert scipy.sparse.issparse(got) and got.shape == (16, 16) and got.dtype == np.float64
    np.testing.assert_array_equal(got.todense(), F)
    F = np.zeros((16, 16))
    F[1, 14] = .4; F[4, 11] = .2; F[11, 1] = .2; F[14, 4] = .2
    np.testing.assert_array_equal(sp.flattening(REF4_SPLITS[1], REF4_TABLE).todense(), F)


def test_reduced_flattening_reference_vectors(sp):
    F = np.array([[0, .4, 0, 0], [.2, 0, 0, 0], [0, 0, 0, .2], [0, 0, .2, 0]])
    np.testing.assert_array_equal(sp.flattening(REF4_SPLITS[0], REF4_TABLE, flattening_format=sp.FlatFormat.reduced), F)
    F = np.array([[0, 0, 0, .4], [0, 0, .2, 0], [.2, 0, 0, 0], [0, .2, 0, 0]])
    np.testing.assert_array_equal(sp.flattening(REF4_SPLITS[1], REF4_TABLE, flattening_format=sp.FlatFormat.reduced), F)


def test_subflattening_reference_identity(sp):
    sel = [3, 7, 11, 12, 13, 14, 15]
    S = np.array([[1, -1], [1, 1]])
    S0 = np.kron(S, S)
    S = np.kron(S0, S0)
    for split in REF4_SPLITS:
        F = np.asarray(sp.flattening(split, REF4_TABLE).todense())
        want = (S @ F @ S.T)[np.ix_(sel, sel)]
        np.testing.assert_allclose(sp.subflattening(split, REF4_TABLE), want, rtol=1e-15, atol=1e-15)


def test_ref4_golden_and_split_forms(sp, golden):
    g = golden("ref4")
    for i, s in enumerate(REF4_SPLITS):
        np.testing.assert_array_equal(np.asarray(sp.flattening(s, REF4_TABLE).todense()), g[f"sparse_{i}"])
        np.testing.assert_array_equal(sp.flattening(s, REF4_TABLE, sp.FlatFormat.reduced), g[f"reduced_{i}"])
        np.testing.assert_allclose(sp.subflattening(s, REF4_TABLE), g[f"subflat_{i}"], rtol=0, atol=1e-15)
        assert sp.split_score(sp.subflattening(s, REF4_TABLE)) ** 2 <= 1e-13   # 7 x 7 of rank 4: degenerate, compare score^2
        assert sp.split_score(sp.flattening(s, REF4_TABLE, sp.FlatFormat.reduced)) == 0.0  # 4 x 4
    # string split, taxa listed in a non-sorted order inside each half
    np.testing.assert_array_equal(np.asarray(sp.flattening("10|32", REF4_TABLE).todense()), g["sparse_str"])
    np.testing.assert_array_equal(sp.flattening("10|32", REF4_TABLE, sp.FlatFormat.reduced), g["reduced_str"])
    np.testing.assert_allclose(sp.subflattening("10|32", REF4_TABLE), g["subflat_str"], rtol=0, atol=1e-15)
    assert sp.flattening("10|32", REF4_TABLE, "something else") is None
    with pytest.raises(KeyError):
        sp.flattening("10|32", {"ATCG": 0.5, "ATCN": 0.5})          # character outside ACGT
    with pytest.raises(KeyError):
        dev4 = sp.DeviceAlignment.from_table(REF4_TABLE, taxa=("0", "1", "2", "3"))
        sp.flattening("10|39", dev4)                                # unknown taxon (table carries .taxa)
    data = {}
    sp.subflattening(REF4_SPLITS[0], REF4_TABLE, data)
    assert set(data) == {"coeffs", "labels"} and data["labels"][2][-1] == "TT"


# ---------------------------------------------------------------- 10 taxa
def _n10(golden, name):
    g = golden(name)
    names = taxa_names(10)
    splits = [mask_to_split(int(m), 10, names) for m in g["masks"]]
    table = O.unpack_table(g["keys"], g["probs"], 10)
    return g, names, splits, table


@pytest.mark.parametrize("name", ["n10_L10k", "n10_L100k"])
def test_n10_api_matrices_bit_exact(sp, golden, name):
    g, names, splits, table = _n10(golden, name)
    dev = sp.DeviceAlignment.from_table(table, taxa=names)
    assert dev.info()["exact"] and dev.info()["N"] in (int(g["L"]), int(g["L"]) // 2, int(g["L"]) // 4)
    for i in g["full_ids"]:
        want = g[f"reduced_{i}"]
        got = sp.flattening(splits[i], dev, sp.FlatFormat.reduced)
        assert got.dtype == np.float64 and got.shape == want.shape
        np.testing.assert_array_equal(got, want)                         # bit-exact vs the reference
        got2 = sp.flattening(splits[i], table, sp.FlatFormat.reduced)    # plain dict input
        np.testing.assert_array_equal(got2, want)
        assert abs(sp.split_score(got) - g["scores"][i]) <= SCORE_TOL
    # sparse format against the reference's stored triplets
    i0 = int(g["sparse_ids"][0])
    S = sp.flattening(splits[i0], dev)
    want = scipy.sparse.coo_matrix((g["sparse0_vals"], (g["sparse0_rows"], g["sparse0_cols"])),
                                   shape=tuple(g["sparse0_shape"]))
    assert S.shape == want.shape and (abs(S.tocoo() - want)).nnz == 0
    for j, i in enumerate(g["sparse_ids"][:3]):
        assert abs(sp.split_score(sp.flattening(splits[int(i)], dev)) - g["sparse_scores"][j]) <= SCORE_TOL
    # subflattening
    big_n = int(g["L"])
    for i in g["sub_ids"]:
        want = g[f"subflat_{i}"]
        got = sp.subflattening(splits[i], dev)
        assert np.array_equal(np.rint(got * big_n), np.rint(want * big_n))
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=5e-14)
        assert abs(sp.split_score(got) - g[f"subscore_{i}"]) <= SCORE_TOL


@pytest.mark.parametrize("name", ["n10_L10k", "n10_L100k"])
def test_n10_batched_scores_all_501(sp, golden, name):
    g, names, splits, table = _n10(golden, name)
    dev = sp.DeviceAlignment.from_table(table, taxa=names)
    scores, status = sp.score_splits(dev, splits, return_status=True)
    assert scores.shape == (501,)
    assert np.all((status & 1) == 0), "eigen iteration hit its cap"
    err = np.abs(scores - g["scores"])
    assert err.max() <= SCORE_TOL, (err.max(), int(err.argmax()))
    # weights-only (fp64 Gram, non-exact) route must agree too
    dev_w = sp.DeviceAlignment.from_arrays(g["keys"], g["probs"], 10, taxa=names, exact=False)
    assert not dev_w.info()["exact"]
    s_w = sp.score_splits(dev_w, splits)
    assert np.abs(s_w - g["scores"]).max() <= SCORE_TOL
    # subflattening method, batched, against per-split reference values
    sub = sp.score_splits(dev, [splits[int(i)] for i in g["sub_ids"]], method=sp.Method.subflattening)
    want = np.array([g[f"subscore_{int(i)}"] for i in g["sub_ids"]])
    assert np.abs(sub - want).max() <= SCORE_TOL


def test_dense_counts_bit_exact(sp, golden):
    g, names, splits, table = _n10(golden, "n10_L100k")
    dev = sp.DeviceAlignment.from_table(table, taxa=names)
    import ctypes as C
    big_n = dev.info()["N"]
    counts = np.rint(g["probs"] * big_n).astype(np.int64)
    for i in (0, 100, 300, 500):
        oa = np.array([names.index(t) for t in splits[i][0]], dtype=np.int32)
        ob = np.array([names.index(t) for t in splits[i][1]], dtype=np.int32)
        out = np.empty(4 ** 10, dtype=np.uint32)
        sp._lib.check(dev.ctx._lib.sp_flatten_dense_counts(
            dev.handle, oa.ctypes.data_as(C.POINTER(C.c_int32)), len(oa), ob.ctypes.data_as(C.POINTER(C.c_int32)),
            len(ob), out.ctypes.data_as(C.POINTER(C.c_uint32))))
        want = O.dense_flattening_packed(g["keys"], counts, 10, oa, ob)
        assert np.array_equal(out.reshape(want.shape), want.astype(np.uint32))
        assert out.sum() == big_n


def test_n16_subflattening(sp, golden):
    g = golden("n16_L4k")
    names = [str(x) for x in g["taxa"]]
    table = O.unpack_table(g["keys"], g["probs"], 16)
    dev = sp.DeviceAlignment.from_table(table, taxa=names)
    big_n = int(g["L"])
    splits, want_scores = [], []
    for i in range(int(g["n_splits"])):
        oa, ob = g[f"orderA_{i}"], g[f"orderB_{i}"]
        split = (tuple(names[t] for t in oa), tuple(names[t] for t in ob))
        got = sp.subflattening(split, dev)
        want = g[f"subflat_{i}"]
        assert np.array_equal(np.rint(got * big_n), np.rint(want * big_n))
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=5e-14)
        splits.append(split)
        want_scores.append(float(g[f"subscore_{i}"]))
    got = sp.score_splits(dev, splits, method=sp.Method.subflattening)
    assert np.abs(got - np.array(want_scores)).max() <= SCORE_TOL


def test_degenerate_and_generic_matrices(sp, golden):
    g = golden("degenerate")
    for k in ("generic", "realvalued"):
        assert abs(sp.split_score(g[k]) - g[k + "_score"]) <= SCORE_TOL
        assert abs(sp.split_score(g[k].T.copy()) - g[k + "_score"]) <= SCORE_TOL
    assert sp.split_score(g["thin3"]) == 0.0
    r = sp.split_score(g["rank4"])          # reference: ~0 or nan (no clamp); here clamped
    assert r * r <= 1e-13
    assert np.isnan(sp.split_score(np.zeros((6, 9))))
    S = scipy.sparse.coo_matrix((g["sp_vals"], (g["sp_rows"], g["sp_cols"])), shape=tuple(g["sp_shape"])).todok()
    assert abs(sp.split_score(S) - g["sp_score"]) <= SCORE_TOL
    assert isinstance(sp.split_score(S), float)
    # randomised differential test against the oracle, ragged shapes incl. min(shape) in {5, 16, 17, 63, 65}
    rng = np.random.default_rng(11)
    for shape in [(5, 40), (40, 5), (16, 16), (17, 300), (63, 64), (65, 130), (200, 31), (257, 300)]:
        M = rng.integers(0, 7, size=shape).astype(np.float64)
        assert abs(sp.split_score(M) - O.dense_split_score(M)) <= SCORE_TOL, shape
    # hard spectrum (no gap after the 4th singular value): slow convergence must still be correct
    M = rng.standard_normal((120, 400))
    assert abs(sp.split_score(M) - O.dense_split_score(M)) <= 1e-9
    # scale invariance far outside the f32 range of the squared Gram values (tools/gpu_fuzz_matrix.py: the score was 1.0)
    M = np.where(rng.random((300, 500)) < 0.1, rng.integers(1, 1000, (300, 500)), 0).astype(np.float64)
    want = O.dense_split_score(M)
    for scale in (1e-8, 1e5, 1e-30, 1e30):
        assert abs(sp.split_score(M * scale) - want) <= SCORE_TOL, scale
        assert abs(sp.split_score(scipy.sparse.csr_matrix(M * scale)) - want) <= SCORE_TOL, scale
    # numerical rank 3 + noise: the 4th singular value sits inside the cluster of noise values, 1e-9 below the first -
    # invisible to the squared iteration; the first-power acceptance has to find it.  1 - top4/trace is ~1e-10 here and
    # good to ~5e-14 (rounding floor of an fp64 Gram matrix), i.e. the score to a few 1e-10
    M = rng.standard_normal((130, 3)) @ rng.standard_normal((3, 257)) + 1e-5 * rng.standard_normal((130, 257))
    want, got = O.dense_split_score(M), sp.split_score(M)
    assert abs(want ** 2 - got ** 2) <= 5e-14 and abs(want - got) <= 5e-9, (want, got)


def test_histogram_from_sequences(sp):
    from splitp_amd import synthetic as syn

    sites = syn.simulate_sites(10, 20000, 0.05, seed=3)
    keys, counts = syn.pattern_table(sites)
    seqs = syn.sequences_ascii(sites)
    # sprinkle invalid characters: those sites must be dropped (fasta.py:54-57), lower case accepted
    seqs = seqs.copy()
    seqs[3, 10] = ord("N"); seqs[0, 11] = ord("-"); seqs[5, 12] = ord("a") if seqs[5, 12] == ord("A") else seqs[5, 12]
    valid = np.ones(20000, dtype=bool); valid[[10, 11]] = False
    k2, c2 = syn.pattern_table(sites[valid])
    dev = sp.DeviceAlignment.from_sequences(seqs)
    gk, gw, gc = dev.fetch()
    assert dev.info()["N"] == int(valid.sum())
    assert np.array_equal(gk, k2) and np.array_equal(gc, c2)
    assert np.array_equal(gw, c2 / float(valid.sum()))
    # the two histogram forms (direct 4^n bins / radix sort + run-length encode) on the same input
    ctx = sp.get_context()
    for force in (1, 0):
        ctx.set_option("hist_sort", force)
        try:
            d3 = sp.DeviceAlignment.from_sequences(seqs)
        finally:
            ctx.set_option("hist_sort", -1)
        fk, fw, fc = d3.fetch()
        assert d3.info()["N"] == int(valid.sum()) and np.array_equal(fk, k2) and np.array_equal(fc, c2)
        assert np.array_equal(fw, gw)
    dev2 = sp.DeviceAlignment.from_site_keys(syn.site_keys(sites), 10)
    gk, gw, gc = dev2.fetch()
    assert np.array_equal(gk, keys) and np.array_equal(gc, counts)
    # reference's FASTA test fixture (tests/test_files/test_alignment_1.fa + tests/test_parsers.py:16-25)
    dev3 = sp.DeviceAlignment.from_sequences(["AAGCT", "TTAGC", "CCTAG", "GGCTA"])
    assert dict(dev3.items()) == {"ATCG": 2 / 5, "CGAT": 1 / 5, "GATC": 1 / 5, "TCGA": 1 / 5}


PROP_TOL = 5e-12   # self-consistency of two converged runs (parity bar vs the oracle stays SCORE_TOL = 1e-10)


def test_full_size_properties(sp):
    """BASELINE config 2 size (10 taxa, 100k bp, all 501 splits) on a fresh synthetic alignment:
    size-independent properties instead of a stored answer."""
    from splitp_amd import synthetic as syn

    sites = syn.simulate_sites(10, 100_000, 0.05, seed=21)
    keys, counts = syn.pattern_table(sites)
    names = taxa_names(10)
    dev = sp.DeviceAlignment.from_arrays(keys, None, 10, counts=counts, n_sites=100_000, taxa=names)
    splits = list(sp.all_splits(names))
    s1 = sp.score_splits(dev, splits)
    # (1) run-to-run bitwise reproducibility (integer Gram, fixed reduction orders)
    s2 = sp.score_splits(dev, splits)
    assert np.array_equal(s1, s2)
    # (2) side swap: score(A|B) == score(B|A) (transpose has the same singular values)
    # (the iteration stops at a relative tail estimate of 1e-13 in the Ritz sum: < 1e-12 in a score, see sparse.hip)
    swapped = [(b, a) for a, b in splits]
    s3 = sp.score_splits(dev, swapped)
    assert np.abs(s1 - s3).max() <= PROP_TOL
    # (3) taxon order inside a half only permutes rows/cols
    perm = [(tuple(reversed(a)), b) for a, b in splits]
    s4 = sp.score_splits(dev, perm)
    assert np.abs(s1 - s4).max() <= PROP_TOL
    # (4) the tree's true splits score lowest
    tree = syn.tree_splits(syn.balanced_tree(10), 10)
    is_true = np.array([frozenset(names.index(t) for t in a) in tree or frozenset(names.index(t) for t in b) in tree
                        for a, b in splits])
    assert is_true.sum() == 7 and s1[is_true].max() < s1[~is_true].min()
    # (5) scale invariance: doubling every count leaves the scores unchanged
    dev2 = sp.DeviceAlignment.from_arrays(keys, None, 10, counts=2 * counts, n_sites=200_000, taxa=names)
    s5 = sp.score_splits(dev2, splits)
    assert np.abs(s1 - s5).max() <= PROP_TOL
    # (6) oracle spot checks at full size
    for i in (0, 77, 250, 480):
        oa = [names.index(t) for t in splits[i][0]]
        ob = [names.index(t) for t in splits[i][1]]
        M = O.reduced_flattening_packed(keys, counts / 100_000.0, 10, oa, ob)[0]
        assert abs(O.dense_split_score(M) - s1[i]) <= SCORE_TOL


def test_gram_kernels_agree_bitwise(sp, golden):
    """int8-limb Gram (default for count tables) vs fp64 MFMA Gram: both are exact for integer counts, so the
    scores must be bit-identical; also covers 1-limb (all counts < 128) and 3-limb (counts >= 16384) tables."""
    from splitp_amd import synthetic as syn

    names = taxa_names(10)
    splits = list(sp.all_splits(names))
    for length, seed in ((100_000, 5), (300, 6), (400_000, 7)):
        sites = syn.simulate_sites(10, length, 0.05, seed=seed)
        keys, counts = syn.pattern_table(sites)
        dev = sp.DeviceAlignment.from_arrays(keys, None, 10, counts=counts, n_sites=length, taxa=names)
        sub = splits[::3]
        dev.ctx.set_gram_mode("auto")
        s_i8 = sp.score_splits(dev, sub, route="dense")
        dev.ctx.set_gram_mode("f64")
        s_f64 = sp.score_splits(dev, sub, route="dense")
        dev.ctx.set_gram_mode("auto")
        assert np.array_equal(s_i8, s_f64), (length, int(counts.max()), np.abs(s_i8 - s_f64).max())
        dev.ctx.set_option("gram_tile64", 1)           # round-1 kernel (64 x 64 tiles) against the 128 x 128-tile kernel
        s_t64 = sp.score_splits(dev, sub, route="dense")
        dev.ctx.set_option("gram_tile64", 0)
        assert np.array_equal(s_i8, s_t64), (length, np.abs(s_i8 - s_t64).max())
        dev.ctx.set_option("eigen_one_stream", 1)      # eigen phase as one pipeline instead of long sides | short sides
        s_one = sp.score_splits(dev, sub, route="dense")
        dev.ctx.set_option("eigen_one_stream", 0)
        assert np.array_equal(s_i8, s_one), (length, np.abs(s_i8 - s_one).max())
        for i in (0, 60, 150):
            a, b = sub[i]
            M = O.reduced_flattening_packed(keys, counts.astype(np.float64), 10, [names.index(t) for t in a],
                                            [names.index(t) for t in b])[0]
            assert abs(O.dense_split_score(M) - s_i8[i]) <= SCORE_TOL


def test_gram_big_tiles_ragged_shapes(sp):
    """128 x 128-tile int8 Gram on shapes whose rows in use are not a multiple of 128 (padding rows are loaded as zeros,
    never read), single-tile short sides, 6 - 10 taxa: bit-identical to the 64 x 64-tile kernel and to the fp64 MFMA
    Gram, and equal to the oracle."""
    from splitp_amd import synthetic as syn

    for n, length, branch, seed in ((6, 3000, 0.08, 11), (8, 20_000, 0.1, 12), (8, 6000, 0.3, 13), (10, 30_000, 0.03, 14),
                                    (10, 2500, 0.3, 15)):
        names = taxa_names(n)
        sites = syn.simulate_sites(n, length, branch, seed=seed)
        keys, counts = syn.pattern_table(sites)
        dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=length, taxa=names)
        splits = list(sp.all_splits(names))
        sub = splits[:: max(1, len(splits) // 90)]
        s_big = sp.score_splits(dev, sub, route="dense")
        dev.ctx.set_option("gram_tile64", 1)
        s_t64 = sp.score_splits(dev, sub, route="dense")
        dev.ctx.set_option("gram_tile64", 0)
        dev.ctx.set_gram_mode("f64")
        s_f64 = sp.score_splits(dev, sub, route="dense")
        dev.ctx.set_gram_mode("auto")
        assert np.array_equal(s_big, s_t64), (n, np.abs(s_big - s_t64).max())
        assert np.array_equal(s_big, s_f64), (n, np.abs(s_big - s_f64).max())
        for i in range(0, len(sub), max(1, len(sub) // 6)):
            a, b = sub[i]
            M = O.reduced_flattening_packed(keys, counts.astype(np.float64), n, [names.index(t) for t in a],
                                            [names.index(t) for t in b])[0]
            assert abs(O.dense_split_score(M) - s_big[i]) <= SCORE_TOL, (n, i)


def test_routes_agree(sp, golden):
    """dense route (scatter + MFMA Gram + 16-wide eigen through HBM), sparse route (one workgroup per split, all in
    LDS, 4-wide block) and auto must give the same scores; sparse twice must be bit-identical (deterministic)."""
    names = taxa_names(10)
    for name in ("n10_L100k", "n10_L10k"):
        g = golden(name)
        splits = [mask_to_split(int(m), 10, names) for m in g["masks"]]
        dev = sp.DeviceAlignment.from_table(O.unpack_table(g["keys"], g["probs"], 10), taxa=names)
        s_dense, st_d = sp.score_splits(dev, splits, route="dense", return_status=True)
        s_sparse, st_s = sp.score_splits(dev, splits, route="sparse", return_status=True)
        s_auto = sp.score_splits(dev, splits)
        assert np.all((st_d & 3) == 0) and np.all((st_s & 3) == 0)
        assert np.abs(s_dense - g["scores"]).max() <= SCORE_TOL
        assert np.abs(s_sparse - g["scores"]).max() <= SCORE_TOL
        assert np.array_equal(s_auto, s_sparse)
        for _ in range(3):
            assert np.array_equal(sp.score_splits(dev, splits, route="sparse"), s_sparse)
    # counts >= 65536 exceed the 16-bit count field of the list entries: such patterns enter as several table rows whose
    # counts add up.  (a) a table that still fits LDS (short branches: few patterns, huge counts) - all size classes,
    # including the small-side Gram where the pieces of one count meet on one row; (b) a 700 k-site table: too many
    # rows for LDS: the workgroups fall through to the global-memory forms of the kernel
    from splitp_amd import synthetic as syn
    sites = syn.simulate_sites(10, 600_000, 0.004, seed=8)
    keys, counts = syn.pattern_table(sites)
    assert counts.max() >= 2 * 65536 and len(keys) < 4000
    small_big = sp.DeviceAlignment.from_arrays(keys, None, 10, counts=counts, n_sites=600_000, taxa=names)
    s_sp, st_sp = sp.score_splits(small_big, splits, route="sparse", return_status=True)
    assert np.all((st_sp & 3) == 0)
    assert np.abs(s_sp - sp.score_splits(small_big, splits, route="dense")).max() <= SCORE_TOL
    for i in (0, 30, 100, 300, 500):
        M = O.reduced_flattening_packed(keys, counts.astype(np.float64), 10, [names.index(t) for t in splits[i][0]],
                                        [names.index(t) for t in splits[i][1]])[0]
        assert abs(O.dense_split_score(M) - s_sp[i]) <= SCORE_TOL
    sites = syn.simulate_sites(10, 700_000, 0.05, seed=8)
    keys, counts = syn.pattern_table(sites)
    assert counts.max() >= 65536
    big = sp.DeviceAlignment.from_arrays(keys, None, 10, counts=counts, n_sites=700_000, taxa=names)
    # (the hand-back chain runs inside the kernel: "sparse" takes such a table too, in the global-memory forms)
    # (25 k table rows, 2|8 splits: the column counters of the all-global form's sort sit in global memory, where the order
    # in which one wave's atomics on a word are served is not fixed - groups are summed in a different order run to run,
    # (and may stop one half product apart), hence the stop rule's tolerance here and not array_equal as for the forms
    # whose counters are in LDS)
    assert np.abs(sp.score_splits(big, splits[:5], route="sparse") - sp.score_splits(big, splits[:5])).max() <= 1e-11
    sb, stb = sp.score_splits(big, splits[::25], return_status=True)
    assert np.all((stb & 3) == 0) and np.all((stb >> 8) <= 8)          # scored by the sparse kernel (HBM form), not the dense route
    assert np.abs(sb - sp.score_splits(big, splits[::25], route="dense")).max() <= SCORE_TOL
    for i, spl in enumerate(splits[::25][:4]):
        M = O.reduced_flattening_packed(keys, counts.astype(np.float64), 10, [names.index(t) for t in spl[0]],
                                        [names.index(t) for t in spl[1]])[0]
        assert abs(O.dense_split_score(M) - sb[i]) <= SCORE_TOL
    # slow-convergence hand-back: a table without a gap behind the 4th singular value (uniform random patterns)
    rng = np.random.default_rng(3)
    rk = np.unique(rng.integers(0, 4 ** 10, size=3000).astype(np.uint64))
    rc = rng.integers(1, 40, size=len(rk)).astype(np.int64)
    flat = sp.DeviceAlignment.from_arrays(rk, None, 10, counts=rc, n_sites=int(rc.sum()), taxa=names)
    sub = splits[::40]
    s_a, st_a = sp.score_splits(flat, sub, return_status=True)
    s_d = sp.score_splits(flat, sub, route="dense")
    assert np.abs(s_a - s_d).max() <= 1e-9
    for i in (0, 5, 12):
        M = O.reduced_flattening_packed(rk, rc.astype(np.float64), 10, [names.index(t) for t in sub[i][0]],
                                        [names.index(t) for t in sub[i][1]])[0]
        assert abs(O.dense_split_score(M) - s_a[i]) <= 1e-9


def test_async_entry_and_handback(sp, golden):
    """sp_score_splits_async writes scores + status to device buffers without a host sync, and the hand-back chain
    runs ON THE DEVICE: once the stream has run no status word carries bit 1 - also on a table where every split needs
    the last resort (the 8-wide block)."""
    import torch
    from splitp_amd import batch, _lib

    g = golden("n10_L100k")
    names = taxa_names(10)
    splits = [mask_to_split(int(m), 10, names) for m in g["masks"]]
    dev = sp.DeviceAlignment.from_table(O.unpack_table(g["keys"], g["probs"], 10), taxa=names)
    taxa_arr, a_arr = batch.encode_splits(splits, dev, 10)
    sc = torch.zeros(501, dtype=torch.float64, device="cuda")
    st = torch.zeros(501, dtype=torch.int32, device="cuda")
    batch.score_encoded_async(dev, taxa_arr, a_arr, _lib.SP_METHOD_FLATTENING, sc.data_ptr(), st.data_ptr())
    torch.cuda.synchronize()
    s_h, st_h = sc.cpu().numpy(), st.cpu().numpy()
    assert not np.any(st_h & 3) and batch.finish_async(dev, taxa_arr, a_arr, s_h, st_h) == 0
    assert np.abs(s_h - g["scores"]).max() <= SCORE_TOL
    # a gapless table (uniform random patterns): no block certifies anything.  The chain runs on the device to its end - no
    # status bit 1 - and FLAGS what it cannot certify (bit 0, an upper estimate: the wide block gives up as soon as its 8
    # Ritz values have settled with the tail behind them still outweighing the 4th); finish_async hands exactly those to the
    # direct solver
    rng = np.random.default_rng(3)
    rk = np.unique(rng.integers(0, 4 ** 10, size=3000).astype(np.uint64))
    rc = rng.integers(1, 40, size=len(rk)).astype(np.int64)
    flat = sp.DeviceAlignment.from_arrays(rk, None, 10, counts=rc, n_sites=int(rc.sum()), taxa=names)
    sub_t, sub_a = taxa_arr[::50], a_arr[::50]
    sc2 = torch.zeros(len(sub_a), dtype=torch.float64, device="cuda")
    st2 = torch.zeros(len(sub_a), dtype=torch.int32, device="cuda")
    batch.score_encoded_async(flat, sub_t, sub_a, _lib.SP_METHOD_FLATTENING, sc2.data_ptr(), st2.data_ptr())
    torch.cuda.synchronize()
    s2, t2 = sc2.cpu().numpy(), st2.cpu().numpy()
    assert not np.any(t2 & 2)
    ref = sp.score_splits(flat, [splits[i] for i in range(0, 501, 50)], route="dense")
    done = (t2 & 1) == 0
    assert np.all(np.abs(s2 - ref)[done] <= 1e-9)
    assert np.all(s2[~done] >= ref[~done] - 1e-9)
    assert batch.finish_async(flat, sub_t, sub_a, s2, t2) == int((~done).sum())
    assert not np.any(t2 & 3) and np.abs(s2 - ref).max() <= 1e-9
    # the synchronous entry point on the same table finishes its flagged splits itself - same scores
    assert np.abs(sp.score_splits(flat, [splits[i] for i in range(0, 501, 50)]) - ref).max() <= 1e-9


def test_config5_one_device_pass(sp):
    """BASELINE config 5 (batch of 32 simulated 12-taxon alignments x all 2035 splits, throughput mode) as ONE call:
    65 120 items through the sparse route's device-side chain (in-LDS kernel -> lists in global memory -> all arrays in
    global memory; a 100 k-site 12-taxon table has ~13.5 k patterns: the 5|7 and 6|6 splits do not fit the LDS form).
    Equals 32 single-alignment calls bit for bit; >= 4 splits per size class of every alignment against the oracle's
    restatement of the reference (reduced flattening + scipy SVD, constructions.py:31-55 + phylogenetics.py:280-300).
    fp64 throughout (config 5's fp32 wording is not built: fp64 is the stricter choice)."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from threadpoolctl import threadpool_limits
    from splitp_amd import batch, _lib
    from splitp_amd import simulation as sim
    from splitp_amd import synthetic as syn

    n, n_al, length = 12, 32, 100_000
    names = taxa_names(n)
    tree = syn.balanced_tree(n)
    devs = []
    for a in range(n_al):
        d = sim.generate_device_alignment(tree, sim.JukesCantor(), length, seed=100 + a, branch_length=0.05)
        d.taxa = tuple(names)
        devs.append(d)
    taxa_arr, a_arr = sp.encode_all_splits(n)
    S = len(a_arr)
    assert S == 2035
    sc = torch.zeros(n_al * S, dtype=torch.float64, device="cuda")
    st = torch.full((n_al * S,), -1, dtype=torch.int32, device="cuda")
    batch.score_encoded_multi_async(devs, taxa_arr, a_arr, sc.data_ptr(), st.data_ptr())
    torch.cuda.synchronize()
    multi, status = sc.cpu().numpy().reshape(n_al, S), st.cpu().numpy().reshape(n_al, S)
    assert not np.any(status & 3) and np.all(status >> 8 >= 1) and np.all(np.isfinite(multi))
    # plan + lane form of the same pass (sp_score_plan_async on a second context) gives the same bits
    plan = batch.SplitPlan(devs[0].ctx, taxa_arr, a_arr, n)
    stream = torch.cuda.Stream()
    from splitp_amd.device import Context
    lane = Context(devs[0].ctx.device, stream=stream.cuda_stream)
    sc_l = torch.zeros(n_al * S, dtype=torch.float64, device="cuda")
    st_l = torch.zeros(n_al * S, dtype=torch.int32, device="cuda")
    batch.score_plan_async(lane, devs, plan, sc_l.data_ptr(), st_l.data_ptr())
    stream.synchronize()
    assert np.array_equal(sc_l.cpu().numpy().reshape(n_al, S), multi)
    # 32 single calls (the synchronous entry point with its own hand-back): bit for bit
    for a in (0, 1, 13, 31):
        one, st1 = batch.score_encoded(devs[a], taxa_arr, a_arr, _lib.SP_METHOD_FLATTENING)
        assert np.array_equal(one, multi[a]) and not np.any(st1 & 3), a
    one_by_one = np.stack([batch.score_encoded(d, taxa_arr, a_arr, _lib.SP_METHOD_FLATTENING)[0] for d in devs])
    assert np.array_equal(one_by_one, multi)
    # oracle: 4 splits per size class (k = 2..6) of every alignment
    k_of = np.minimum(a_arr, n - a_arr)
    rng = np.random.default_rng(5)
    jobs = []
    for a, d in enumerate(devs):
        keys, w, cnt = d.fetch()
        for k in range(2, 7):
            for i in rng.choice(np.nonzero(k_of == k)[0], size=4, replace=False):
                jobs.append((a, int(i), keys, cnt))

    def check(job):
        a, i, keys, cnt = job
        m = O.reduced_flattening_packed(keys, cnt.astype(np.float64), n, list(taxa_arr[i, :a_arr[i]]),
                                        list(taxa_arr[i, a_arr[i]:]))[0]
        return abs(O.dense_split_score(m) - multi[a, i])

    with threadpool_limits(limits=4):
        with ThreadPoolExecutor(4) as pool:
            errs = list(pool.map(check, jobs))
    assert len(errs) == n_al * 20 and max(errs) <= SCORE_TOL, max(errs)


def test_lanes_share_plan_and_alignment(sp, golden):
    """Three lane contexts (one HIP stream each) score the same alignment with one shared immutable plan, several calls
    in flight: every result equals the single-stream result bit for bit (sparse route), and the dense route driven
    through per-lane ALIGNMENTS' own contexts is unaffected by the neighbours (the old unordered-lane API
    raced on the dense route's pools - it is gone)."""
    import torch
    from splitp_amd import batch, _lib
    from splitp_amd.device import Context

    g = golden("n10_L100k")
    names = taxa_names(10)
    splits = [mask_to_split(int(m), 10, names) for m in g["masks"]]
    dev = sp.DeviceAlignment.from_table(O.unpack_table(g["keys"], g["probs"], 10), taxa=names)
    taxa_arr, a_arr = batch.encode_splits(splits, dev, 10)
    base, st0 = batch.score_encoded(dev, taxa_arr, a_arr, _lib.SP_METHOD_FLATTENING_SPARSE)
    plan = batch.SplitPlan(dev.ctx, taxa_arr, a_arr, 10)
    streams = [torch.cuda.Stream() for _ in range(3)]
    lanes = [Context(dev.ctx.device, stream=s.cuda_stream) for s in streams]
    outs = [[(torch.zeros(501, dtype=torch.float64, device="cuda"), torch.zeros(501, dtype=torch.int32, device="cuda"))
             for _ in range(4)] for _ in lanes]
    for rep in range(4):
        for li, lane in enumerate(lanes):
            sc, st = outs[li][rep]
            batch.score_plan_async(lane, [dev], plan, sc.data_ptr(), st.data_ptr())
    torch.cuda.synchronize()
    for li in range(3):
        for rep in range(4):
            assert np.array_equal(outs[li][rep][0].cpu().numpy(), base), (li, rep)
            assert not np.any(outs[li][rep][1].cpu().numpy() & 3)
    assert np.abs(base - g["scores"]).max() <= SCORE_TOL
    # dense route: asynchronous entry point on the alignment's own context, back to back - ordered on one stream
    dsc = [torch.zeros(501, dtype=torch.float64, device="cuda") for _ in range(3)]
    dst = [torch.zeros(501, dtype=torch.int32, device="cuda") for _ in range(3)]
    for i in range(3):
        batch.score_encoded_async(dev, taxa_arr, a_arr, _lib.SP_METHOD_FLATTENING_DENSE, dsc[i].data_ptr(), dst[i].data_ptr())
    torch.cuda.synchronize()
    d0 = dsc[0].cpu().numpy()
    assert np.array_equal(d0, dsc[1].cpu().numpy()) and np.array_equal(d0, dsc[2].cpu().numpy())
    assert np.abs(d0 - g["scores"]).max() <= SCORE_TOL


def test_enoconv_is_reported(sp):
    """A split that exhausts the whole chain keeps its estimate and is REPORTED: the C ABI returns SP_ENOCONV (5) with
    scores and status written (status bit 0), the Python layer turns that into a RuntimeWarning.  Forced here by capping
    the last resort's half products (option wide_cap) on a 12-taxon table whose splits need the 8-wide block
    (test_wide_block_fallback_12_taxa's table)."""
    import ctypes as C
    import warnings
    from splitp_amd import _lib, batch

    rng = np.random.default_rng(1)
    n = 12
    keys, counts = _copy_mutate_table(rng, n, 20000, 3)
    names = taxa_names(n)
    dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=int(counts.sum()), taxa=names)
    splits = []
    for _ in range(16):
        k = int(rng.integers(2, n - 1))
        left = sorted(rng.choice(n, size=k, replace=False).tolist())
        splits.append((tuple(names[t] for t in left), tuple(names[t] for t in range(n) if t not in left)))
    taxa_arr, a_arr = batch.encode_splits(splits, dev, n)
    good, st_good = batch.score_encoded(dev, taxa_arr, a_arr, _lib.SP_METHOD_FLATTENING)
    wide = np.nonzero((st_good >> 8) > 41)[0]
    assert len(wide) > 0 and not np.any(st_good & 3)
    t2, a2 = np.ascontiguousarray(taxa_arr[wide]), np.ascontiguousarray(a_arr[wide])
    dev.ctx.set_option("wide_cap", 6)
    assert dev.ctx.get_option("wide_cap") == 6
    dev.ctx.set_option("direct_finish", 0)      # (with the direct solver on - the default - nothing stays flagged: below)
    scores = np.zeros(len(wide))
    status = np.zeros(len(wide), dtype=np.int32)
    rc = dev.ctx._lib.sp_score_splits(dev.handle, _lib._ptr(t2, C.c_int32), _lib._ptr(a2, C.c_int32), len(wide),
                                      _lib.SP_METHOD_FLATTENING, _lib._ptr(scores, C.c_double), None,
                                      _lib._ptr(status, C.c_int32))
    assert rc == _lib.SP_ENOCONV and np.all(status & 1) and not np.any(status & 2)
    assert b"without a certificate" in dev.ctx._lib.sp_last_error()
    assert np.all(np.isfinite(scores)) and np.all(scores >= good[wide] - 1e-9)      # upper estimates of the scores
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        est, st = batch.score_encoded(dev, t2, a2, _lib.SP_METHOD_FLATTENING)
    assert any(issubclass(w.category, RuntimeWarning) for w in caught) and np.all(st & 1) and np.array_equal(est, scores)
    dev.ctx.set_option("direct_finish", 1)      # same cap, direct solver on: every flagged split is finished, no warning
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        fin, st_fin = batch.score_encoded(dev, t2, a2, _lib.SP_METHOD_FLATTENING)
    assert not any(issubclass(w.category, RuntimeWarning) for w in caught)
    assert np.all((st_fin & 7) == 4) and np.abs(fin - good[wide]).max() <= SCORE_TOL
    dev.ctx.set_option("wide_cap", 0)
    back, st_back = batch.score_encoded(dev, t2, a2, _lib.SP_METHOD_FLATTENING)
    assert np.array_equal(back, good[wide]) and not np.any(st_back & 3)


def test_multi_alignment_launch_matches_single(sp, golden):
    """sp_score_splits_multi_async: three alignments of different pattern counts in one launch give bit-identical
    scores to three single-alignment calls, and match the golden / oracle scores."""
    import torch
    from splitp_amd import batch, _lib
    from splitp_amd import synthetic as syn

    names = taxa_names(10)
    g = golden("n10_L100k")
    splits = [mask_to_split(int(m), 10, names) for m in g["masks"]]
    devs = [sp.DeviceAlignment.from_table(O.unpack_table(g["keys"], g["probs"], 10), taxa=names)]
    tables = []
    for seed, length in ((11, 20_000), (12, 60_000)):
        keys, counts = syn.pattern_table(syn.simulate_sites(10, length, 0.05, seed=seed))
        devs.append(sp.DeviceAlignment.from_arrays(keys, None, 10, counts=counts, n_sites=length, taxa=names))
        tables.append(syn.table_as_dict(keys, counts, 10, total=length))
    taxa_arr, a_arr = batch.encode_splits(splits, devs[0], 10)
    sc = torch.zeros(3 * 501, dtype=torch.float64, device="cuda")
    st = torch.zeros(3 * 501, dtype=torch.int32, device="cuda")
    batch.score_encoded_multi_async(devs, taxa_arr, a_arr, sc.data_ptr(), st.data_ptr())
    torch.cuda.synchronize()
    multi = sc.cpu().numpy().reshape(3, 501)
    assert (st.cpu().numpy() & 2).sum() == 0
    for i, dev in enumerate(devs):
        one = torch.zeros(501, dtype=torch.float64, device="cuda")
        st1 = torch.zeros(501, dtype=torch.int32, device="cuda")
        batch.score_encoded_async(dev, taxa_arr, a_arr, _lib.SP_METHOD_FLATTENING_SPARSE, one.data_ptr(), st1.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(one.cpu().numpy(), multi[i])
    assert np.abs(multi[0] - g["scores"]).max() <= SCORE_TOL
    for i in (0, 137, 500):
        assert abs(O.split_score(O.flattening(splits[i], tables[1], "reduced")) - multi[2][i]) <= SCORE_TOL


def test_config3_size_subflattening(sp):
    """BASELINE config 3 size (16 taxa, 1M bp, subflattening route): exact moment identity and oracle spot checks."""
    from splitp_amd import synthetic as syn

    n, length = 16, 1_000_000
    sites = syn.simulate_sites(n, length, 0.05, seed=3)
    keys, counts = syn.pattern_table(sites)
    names = syn.taxa_names(n)
    dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=length, taxa=names)
    # ALL 32 751 splits (BASELINE config 3 says "all splits"), enumerated on the device, against the oracle on at least
    # four splits of every size class - first, last and two inside - and against the list-based call on a sample
    got, st = sp.score_all_splits(dev, method=sp.Method.subflattening, return_status=True)
    assert len(got) == 32751 and not np.any(st & 3) and np.all(np.isfinite(got))
    taxa_arr, a_arr = sp.encode_all_splits(n)
    M = O.moment_matrix(keys, counts, n)
    checked = 0
    for k in range(2, 9):
        idx = np.nonzero(np.minimum(a_arr, n - a_arr) == k)[0]          # (the side holding taxon 0 comes first: a = k or n - k)
        for i in {int(idx[0]), int(idx[len(idx) // 3]), int(idx[2 * len(idx) // 3]), int(idx[-1])}:
            oa, ob = taxa_arr[i][:a_arr[i]].tolist(), taxa_arr[i][a_arr[i]:].tolist()
            S = M[np.ix_(O.subflattening_index(oa, n), O.subflattening_index(ob, n))] / float(length)
            assert abs(O.dense_split_score(S) - got[i]) <= SCORE_TOL, (k, i)
            checked += 1
            if i == int(idx[0]):
                split = (tuple(names[t] for t in oa), tuple(names[t] for t in ob))
                m_gpu = sp.subflattening(split, dev)
                assert np.array_equal(np.rint(m_gpu * length), np.rint(S * length))
    assert checked >= 28
    some = list(sp.all_splits(names, size=2))[:40] + list(sp.all_splits(names, size=8))[:2000:50]
    by_list = sp.score_splits(dev, some, method=sp.Method.subflattening)
    assert np.array_equal(by_list[:40], got[:40])          # all_splits order: the size-2 class comes first
    pos8 = np.nonzero(np.minimum(a_arr, n - a_arr) == 8)[0][:2000:50]
    assert np.array_equal(by_list[40:], got[pos8])


def test_erickson_svd_matches_reference(sp, golden):
    """The in-tree caller of the path (reference splitp/phylogenetics.py:99-171), every round one batched device call:
    same chosen splits, in the same order, as the real reference on the 10-taxon 10k-bp golden alignment."""
    import json, os
    from tests.conftest import GOLDEN

    g = golden("n10_L10k")
    table = O.unpack_table(g["keys"], g["probs"], 10)
    want = json.load(open(os.path.join(GOLDEN, "erickson_n10_L10k.json")))
    for method in ("flattening", "subflattening"):
        got = sp.erickson_SVD(table, method=sp.Method[method])
        exp = [tuple(tuple(side) for side in s) for s in want[method]]
        # the first n-3 rounds are decided by real score differences and must match exactly; the last round only
        # re-selects one of the bipartitions already chosen, and which one is a tie between A|B and B|A scored as
        # transposes of each other (equal in exact arithmetic, ~1e-16 apart in floating point in BOTH implementations)
        assert got[:-1] == exp[:-1], (method, got, exp)
        assert len(got) == len(exp) == 8 and got[-1] in got[:-1]
    with pytest.raises(NotImplementedError):
        sp.erickson_SVD(table, method=sp.Method.distance)


def test_mutual_information_score(sp, golden):
    """flattening_rank_1_approximation_divergence (reference splitp/phylogenetics.py:364-373) - the matrix form, the
    batched Method.mutual_information form (count table and float-weight table) and the erickson_SVD that uses it,
    against values / the tree produced by the real reference (tests/golden/divergence_n10_L10k.json)."""
    import json, os
    from tests.conftest import GOLDEN

    want = json.load(open(os.path.join(GOLDEN, "divergence_n10_L10k.json")))
    g = golden("n10_L10k")
    names = taxa_names(10)
    table = O.unpack_table(g["keys"], g["probs"], 10)
    splits = [mask_to_split(int(g["masks"][i]), 10, names) for i in want["split_ids"]]
    ref = np.array(want["divergence"])
    # matrix form on the drop-in reduced flattening
    for j in (0, 7, 20, 41):
        flat = sp.flattening(splits[j], table, sp.FlatFormat.reduced)
        got = sp.phylogenetics.flattening_rank_1_approximation_divergence(flat)
        assert isinstance(got, np.float64) and abs(got - ref[j]) <= SCORE_TOL
    for s, v in zip(REF4_SPLITS, want["ref4"]):
        flat = sp.flattening(s, REF4_TABLE, sp.FlatFormat.reduced)
        assert abs(sp.phylogenetics.flattening_rank_1_approximation_divergence(flat) - v) <= 1e-12
    # batched, count table (integer marginals) and float-weight table (fp64 atomics)
    dev = sp.DeviceAlignment.from_table(table, taxa=names)
    got = sp.score_splits(dev, splits, method=sp.Method.mutual_information)
    assert np.abs(got - ref).max() <= SCORE_TOL
    assert np.array_equal(got, sp.score_splits(dev, splits, method=sp.Method.mutual_information))   # integer marginals: repeatable
    dev_w = sp.DeviceAlignment.from_arrays(g["keys"], g["probs"], 10, taxa=names, exact=False)
    assert np.abs(sp.score_splits(dev_w, splits, method=sp.Method.mutual_information) - ref).max() <= SCORE_TOL
    # oracle (vectorised) on all 501 splits of the 100 k-bp table
    g2 = golden("n10_L100k")
    all_splits_ = [mask_to_split(int(m), 10, names) for m in g2["masks"]]
    dev2 = sp.DeviceAlignment.from_table(O.unpack_table(g2["keys"], g2["probs"], 10), taxa=names)
    got2 = sp.score_splits(dev2, all_splits_, method=sp.Method.mutual_information)
    for i in range(0, 501, 20):
        oa = [names.index(t) for t in all_splits_[i][0]]
        ob = [names.index(t) for t in all_splits_[i][1]]
        assert abs(O.rank1_divergence_packed(g2["keys"], g2["probs"], 10, oa, ob) - got2[i]) <= SCORE_TOL
    # the fused kernel (marginals in LDS) against the global-memory form it replaced for count tables: identical sums
    sp.get_context().set_option("divergence_global", 1)
    try:
        glob2 = sp.score_splits(dev2, all_splits_, method=sp.Method.mutual_information)
    finally:
        sp.get_context().set_option("divergence_global", 0)
    assert np.array_equal(glob2, got2)
    # the caller: neighbour joining by mutual information
    tree = sp.erickson_SVD(table, method=sp.Method.mutual_information)
    exp = [tuple(tuple(side) for side in s) for s in want["erickson_mutual_information"]]
    assert tree[:-1] == exp[:-1] and len(tree) == len(exp) and tree[-1] in tree[:-1]


def test_device_simulator(sp):
    """generate_alignment on the device (reference splitp/simulation.py:9-56): sampling semantics with deterministic
    matrices (column orientation, root uniform), the exact pattern distribution (chi-square against the oracle, JC and
    asymmetric matrices), reproducibility, the reference's Phylogeny duck type, and the scores of a simulated alignment."""
    import networkx as nx
    from splitp_amd import simulation as sim
    from splitp_amd import synthetic as syn

    # (1) deterministic walk: f = 0->1, 1->3, 2->0, 3->2 (not an involution: M != M^T), M[new, old]
    f = [1, 3, 0, 2]
    M = np.zeros((4, 4))
    for old, new in enumerate(f):
        M[new, old] = 1.0

    class Fixed:
        def transition_matrix(self, t):
            return M

    tree4 = ((0, 1), (2, 3))      # every leaf two branches below the root
    table = sim.generate_alignment(tree4, Fixed(), 40_000, seed=1, branch_length=1.0)
    want = {"".join("ACGT"[f[f[r]]] * 4): None for r in range(4)}
    assert set(table) == set(want) and list(table) == sorted(table)
    assert abs(sum(table.values()) - 1) < 1e-12 and all(abs(v - 0.25) < 0.02 for v in table.values())
    # (2) exact distribution: Jukes-Cantor and per-branch asymmetric matrices
    jc = sim.JukesCantor()
    parent, leaf, trans, taxa = sim.tree_arrays(tree4, jc, 0.15)
    n_sites = 2_000_000
    dev = sim.generate_device_alignment(tree4, jc, n_sites, seed=7, branch_length=0.15)
    keys, w, cnt = dev.fetch()
    assert dev.info()["N"] == n_sites and int(cnt.sum()) == n_sites and np.all(np.diff(keys.astype(np.int64)) > 0)
    p = O.pattern_distribution(parent.tolist(), leaf.tolist(), trans, 4)
    freq = np.zeros(256)
    freq[keys.astype(np.int64)] = cnt / n_sites
    chi2 = float(np.sum((freq - p) ** 2 / p) * n_sites)
    assert 255 - 6 * 22.6 < chi2 < 255 + 6 * 22.6, chi2
    rng = np.random.default_rng(11)
    g = nx.DiGraph()
    g.add_edges_from([("r", "x"), ("r", "y"), ("x", "0"), ("x", "1"), ("y", "2"), ("y", "z"), ("z", "3"), ("z", "4")])
    for node in g.nodes:
        m = rng.random((4, 4)) + 2 * np.eye(4)
        g.nodes[node]["transition_matrix"] = m / m.sum(axis=0, keepdims=True)
        g.nodes[node]["branch_length"] = 0.1

    class Phylo:          # the reference's Phylogeny, as far as the simulator looks at it
        networkx_graph = g
        taxa = ["0", "1", "2", "3", "4"]

    parent, leaf, trans, taxa = sim.tree_arrays(Phylo(), None)
    dev5 = sim.generate_device_alignment(Phylo(), None, n_sites, seed=3)
    keys, w, cnt = dev5.fetch()
    p = O.pattern_distribution(parent.tolist(), leaf.tolist(), trans, 5)
    freq = np.zeros(1024)
    freq[keys.astype(np.int64)] = cnt / n_sites
    chi2 = float(np.sum((freq - p) ** 2 / p) * n_sites)
    assert 1023 - 6 * 45.2 < chi2 < 1023 + 6 * 45.2, chi2
    # the same tree through a model object: branch lengths from the nodes
    dev5b = sim.generate_device_alignment(Phylo(), jc, 100_000, seed=3)
    assert dev5b.info()["N"] == 100_000
    # (3) reproducible, seed-dependent
    a = sim.generate_alignment(tree4, jc, 50_000, seed=42, branch_length=0.15)
    b = sim.generate_alignment(tree4, jc, 50_000, seed=42, branch_length=0.15)
    c = sim.generate_alignment(tree4, jc, 50_000, seed=43, branch_length=0.15)
    assert a == b and a != c
    # (4) downstream: a simulated 10-taxon alignment, the tree's own splits score lowest
    names = taxa_names(10)
    tree10 = syn.balanced_tree(10)
    dev10 = sim.generate_device_alignment(tree10, jc, 100_000, seed=9, branch_length=0.05)
    dev10.taxa = tuple(names)
    splits = list(sp.all_splits(names))
    s = sp.score_splits(dev10, splits)
    true = syn.tree_splits(tree10, 10)
    is_true = np.array([frozenset(names.index(t) for t in x) in true or frozenset(names.index(t) for t in y) in true
                        for x, y in splits])
    assert is_true.sum() == 7 and s[is_true].max() < s[~is_true].min()


def test_device_simulator_against_reference_probabilities(sp, golden):
    """The device simulator against the REFERENCE's exact pattern probabilities (simulation.py:58-83
    get_pattern_probabilities with GTR.JukesCantor; fixture tests/golden/extras.npz made by tools/make_goldens.py): the
    reference's own 4-taxon balanced tree and an unbalanced 5-taxon tree with unequal branches, laid out as the
    parents-first arrays sp_simulate_alignment takes, chi-square over all 4^n cells at 2e6 sites."""
    import ctypes as C
    from splitp_amd import _lib
    from splitp_amd.device import DeviceAlignment

    e = golden("extras")
    ctx = sp.get_context()
    n_sites = 2_000_000
    for tag, seed in (("sim4", 21), ("sim5", 22)):
        n = int(e[f"{tag}_n"])
        parent = np.ascontiguousarray(e[f"{tag}_parent"], dtype=np.int32)
        leaf = np.ascontiguousarray(e[f"{tag}_leaf"], dtype=np.int32)
        trans = np.ascontiguousarray(e[f"{tag}_trans"], dtype=np.float64)
        h = C.c_void_p()
        _lib.check(ctx._lib.sp_simulate_alignment(ctx.handle, len(parent), _lib._ptr(parent, C.c_int32),
                                                  _lib._ptr(leaf, C.c_int32), _lib._ptr(trans, C.c_double), n, n_sites,
                                                  C.c_uint64(seed), C.byref(h)))
        dev = DeviceAlignment(h, ctx, n)
        keys, w, cnt = dev.fetch()
        assert int(cnt.sum()) == n_sites
        p = e[f"{tag}_probs"]
        freq = np.zeros(4 ** n)
        freq[keys.astype(np.int64)] = cnt / n_sites
        cells = 4 ** n
        chi2 = float(np.sum((freq - p) ** 2 / p) * n_sites)
        assert cells - 1 - 6 * np.sqrt(2 * (cells - 1)) < chi2 < cells - 1 + 6 * np.sqrt(2 * (cells - 1)), (tag, chi2)


def test_banned_patterns_and_rank_k_approximation(sp, golden):
    """sparse_flattening_with_banned_patterns (constructions.py:94-105) and flattening_rank_k_approximation
    (phylogenetics.py:343-361) on the device-resident table against the real reference's output (extras.npz): every
    letter on either side of a 4-taxon table, three letter / side combinations and the rank-k matrix of a 10-taxon split."""
    from splitp_amd.phylogenetics import flattening_rank_1_approximation, flattening_rank_k_approximation

    e = golden("extras")
    t7 = O.unpack_table(e["t7_keys"], e["t7_probs"], 4)
    taxa4 = ["0", "1", "2", "3"]
    split4 = (("0", "1"), ("2", "3"))
    for ch in "ACGT":
        for side in ("row", "col"):
            m = sp.sparse_flattening_with_banned_patterns(split4, t7, taxa4, **{f"ban_{side}_patterns": ch})
            assert scipy_is_dok(m) and m.shape == (16, 16)
            coo = m.tocoo()
            want = {(int(r), int(c)): v for r, c, v in zip(e[f"t7_ban_{side}_{ch}_r"], e[f"t7_ban_{side}_{ch}_c"],
                                                          e[f"t7_ban_{side}_{ch}_v"])}
            assert {(int(r), int(c)): v for r, c, v in zip(coo.row, coo.col, coo.data)} == want, (ch, side)
    rk4 = flattening_rank_k_approximation(split4, t7)
    assert rk4.shape == (16, 16) and np.abs(np.asarray(rk4.todense()) - e["t7_rank_k"]).max() <= 1e-16
    g = golden("n10_L10k")
    names = taxa_names(10)
    table = O.unpack_table(g["keys"], g["probs"], 10)
    dev = sp.DeviceAlignment.from_table(table, taxa=names)
    split = mask_to_split(int(g["masks"][int(e["n10_split_id"])]), 10, names)
    for ch, side in (("A", "row"), ("T", "col"), ("G", "row")):
        coo = sp.sparse_flattening_with_banned_patterns(split, dev, names, **{f"ban_{side}_patterns": ch}).tocoo()
        order = np.lexsort((coo.col, coo.row))
        assert np.array_equal(coo.row[order], e[f"n10_ban_{side}_{ch}_r"])
        assert np.array_equal(coo.col[order], e[f"n10_ban_{side}_{ch}_c"])
        assert np.array_equal(coo.data[order], e[f"n10_ban_{side}_{ch}_v"])
    rk = flattening_rank_k_approximation(split, table).tocsr()
    assert tuple(rk.shape) == tuple(e["n10_rank_k_shape"]) and rk.nnz == int(e["n10_rank_k_nnz"])
    assert abs(rk.sum() - float(e["n10_rank_k_sum"])) <= 1e-14
    assert abs(np.sqrt(rk.multiply(rk).sum()) - float(e["n10_rank_k_fro"])) <= 1e-14
    got = np.asarray(rk[e["n10_rank_k_r"], e["n10_rank_k_c"]]).ravel()
    assert np.abs(got - e["n10_rank_k_v"]).max() <= 1e-17
    # rank-1 approximation of a reduced flattening: product of the marginals, the reference's three return forms
    F = sp.flattening(split, dev, sp.FlatFormat.reduced)
    approx, r, c = flattening_rank_1_approximation(F, return_vectors=True)
    assert approx.shape == F.shape[::-1] and np.allclose(r, F.sum(axis=0)) and np.allclose(c, F.sum(axis=1))
    assert flattening_rank_1_approximation(F, dont_compute_matrix=True) is None
    # frobenius_norm on the drop-in outputs (matrix.py:7-14)
    assert abs(sp.frobenius_norm(sp.flattening(split, dev)) - float(e["fro_sparse"])) <= 2e-14
    assert abs(sp.frobenius_norm(F) - float(e["fro_dense"])) <= 2e-14


def scipy_is_dok(m):
    from scipy.sparse import dok_matrix

    return isinstance(m, dok_matrix)


def test_dropin_loop_prefetched_scores(sp, golden):
    """Round 4 (re-entry): once split_score has consumed an untouched Flattening of a table, the next flattening() of that
    table enqueues the score of its split behind the fetch (constructions._ScorePrefetch).  Same kernels on the same resident table: the scores equal the batched call's bit for bit, with the prefetch on,
    off, overtaken (more matrices alive than ring slots) and dropped (matrix edited in place); oracle on top."""
    from splitp_amd import constructions as K, device
    from splitp_amd.constructions import Flattening, flattening_origin

    g = golden("n10_L10k")
    names = taxa_names(10)
    table = O.unpack_table(g["keys"], g["probs"], 10)
    splits = [mask_to_split(int(m), 10, names) for m in g["masks"]]
    pick = list(range(0, 501, 7))
    batched = sp.score_splits(table, [splits[i] for i in pick])
    assert np.max(np.abs(batched - g["scores"][pick])) <= SCORE_TOL
    device.clear_table_cache()
    al = device.as_device_alignment(table)
    assert K.PREFETCH_SCORES

    def loop():
        out, used = [], 0
        for i in pick:
            F = sp.flattening(splits[i], table, sp.FlatFormat.reduced)
            assert type(F) is Flattening and F.flags.writeable and F.flags.c_contiguous and F.dtype == np.float64
            used += F._sp_pending is not None
            s = sp.split_score(F)
            assert isinstance(s, np.float64) and F._sp_pending is None
            assert sp.split_score(F) == s                      # asked again: the synchronous call, the same kernels
            out.append(s)
        return np.array(out), used

    before = K._prefetchers[al.ctx.device].issued if al.ctx.device in K._prefetchers else 0
    got, used = loop()
    assert np.array_equal(got, batched)
    assert used == len(pick) - 1, used                          # every flattening after the first scored one was prefetched
    pf = K._prefetchers[al.ctx.device]
    assert pf.issued - before == used
    for fid in g["full_ids"]:                                   # np.empty + fetch: the reference's matrix, bit for bit
        F = sp.flattening(splits[int(fid)], table, sp.FlatFormat.reduced)
        assert np.array_equal(np.asarray(F), g[f"reduced_{int(fid)}"])
    # a caller who only takes flattenings: the credit of two runs out, nothing further is enqueued
    n0 = pf.issued
    keep = [sp.flattening(splits[i], table, sp.FlatFormat.reduced) for i in pick[:6]]
    assert pf.issued - n0 <= 2 and sum(F._sp_pending is not None for F in keep) == pf.issued - n0
    assert np.array_equal(np.array([sp.split_score(F) for F in keep]), batched[:6])
    # more prefetched matrices alive than ring slots: the overtaken ones are scored by the synchronous call
    many = []
    for j in range(K._PREFETCH_SLOTS + 8):
        al._sp_prefetch_credit = 2
        many.append(sp.flattening(splits[pick[j % len(pick)]], table, sp.FlatFormat.reduced))
    assert all(F._sp_pending is not None for F in many)
    sc = np.array([sp.split_score(F) for F in many])
    assert np.array_equal(sc, batched[[j % len(pick) for j in range(len(many))]])
    # edited in place after the prefetch: the pending score is dropped with the origin, the matrix is scored as it now is
    al._sp_prefetch_credit = 2
    F = sp.flattening(splits[pick[40]], table, sp.FlatFormat.reduced)
    assert F._sp_pending is not None
    F[0, 0] += 0.5
    edited = sp.split_score(F)
    assert flattening_origin(F) is None and F._sp_pending is None
    assert abs(edited - O.dense_split_score(np.asarray(F))) <= SCORE_TOL and abs(edited - batched[40]) > 1e-6
    # switched off: the round-3 behaviour, the same numbers
    K.PREFETCH_SCORES = False
    try:
        n0 = pf.issued
        got2, used2 = loop()
        assert used2 == 0 and pf.issued == n0 and np.array_equal(got2, batched)
    finally:
        K.PREFETCH_SCORES = True
    # a float-weight table (no integer counts behind the values: the dense route's asynchronous entry) - prefetched and
    # synchronous scores agree bit for bit, and with the oracle
    rng = np.random.default_rng(77)
    wtable = {k: float(v) * float(rng.uniform(0.5, 1.5)) for k, v in table.items()}
    few = pick[::9]
    res = {}
    for on in (True, False):
        K.PREFETCH_SCORES = on
        device.clear_table_cache()
        out, used = [], 0
        for i in few:
            F = sp.flattening(splits[i], wtable, sp.FlatFormat.reduced)
            used += F._sp_pending is not None
            keepF = np.array(F)
            out.append(sp.split_score(F))
            assert abs(out[-1] - O.dense_split_score(keepF)) <= SCORE_TOL
        assert used == (len(few) - 1 if on else 0)
        res[on] = np.array(out)
    K.PREFETCH_SCORES = True
    assert np.array_equal(res[True], res[False])


def test_dropin_loop_stays_on_the_device(sp, golden):
    """The unchanged README loop (README.md:37-41) on a plain dict: the table is uploaded once (content-checked cache),
    flattening(..., reduced) returns the reference's ndarray (bit-exact) that remembers its origin, and split_score(F)
    scores the split from the resident table - same scores as the batched call, no second upload.  Anything derived
    from F is a plain array and takes the generic matrix route."""
    import pickle
    from splitp_amd import device
    from splitp_amd.constructions import Flattening, flattening_origin

    g = golden("n10_L10k")
    names = taxa_names(10)
    table = O.unpack_table(g["keys"], g["probs"], 10)          # plain dict, no .taxa: sorted(union) resolves the names
    splits = [mask_to_split(int(m), 10, names) for m in g["masks"]]
    device.clear_table_cache()
    al1 = device.as_device_alignment(table)
    assert device.as_device_alignment(table) is al1             # hit: same object, same content
    scores = []
    for i in range(0, 501, 25):
        F = sp.flattening(splits[i], table, sp.FlatFormat.reduced)
        assert type(F) is Flattening and isinstance(F, np.ndarray) and F.dtype == np.float64 and F.flags.writeable
        assert flattening_origin(F) is not None and flattening_origin(F)[0] is al1
        s = sp.split_score(F)
        assert isinstance(s, np.float64) and abs(s - g["scores"][i]) <= SCORE_TOL
        scores.append(s)
        if i in (0, 250, 500):
            full = int(g["full_ids"][0])
            plain = F.copy()                                    # ordinary writable ndarray: the generic dense route
            assert type(plain) is np.ndarray or flattening_origin(plain) is None
            assert plain.flags.writeable and abs(sp.split_score(plain) - s) <= SCORE_TOL
            assert flattening_origin(F * 1.0) is None and flattening_origin(F[:, :]) is None
            assert flattening_origin(pickle.loads(pickle.dumps(F))) is None
            # the reference's array is writable (constructions.py:51): an in-place edit is allowed, and the edited array
            # is scored as the matrix it now is (generic route), not as the split it came from
            keep = F[0, 0]
            F[0, 0] = keep + 0.5
            assert flattening_origin(F) is None
            edited = sp.split_score(F)
            assert abs(edited - O.dense_split_score(np.asarray(F))) <= SCORE_TOL and abs(edited - s) > 1e-6
            G = sp.flattening(splits[i], table, sp.FlatFormat.reduced)
            G *= 3.0                                            # an in-place rescale: same score (scale invariant), but
            assert flattening_origin(G) is None                 # a different matrix: generic route
            assert abs(sp.split_score(G) - s) <= SCORE_TOL
            G2 = sp.flattening(splits[i], table, sp.FlatFormat.reduced)
            G2 /= 1.0                                           # an in-place operation that changes nothing keeps the origin
            assert flattening_origin(G2) is not None
    for fid in g["full_ids"]:
        F = sp.flattening(splits[int(fid)], table, sp.FlatFormat.reduced)
        assert np.array_equal(np.asarray(F), g[f"reduced_{int(fid)}"])
    batched = sp.score_splits(table, splits[0:501:25])
    assert np.array_equal(np.array(scores), batched)            # the same kernels on the same resident table
    # a table edited in place is a different table: miss, fresh upload, different scores
    some = next(iter(table))
    table[some] = table[some] * 2.0
    al2 = device.as_device_alignment(table)
    assert al2 is not al1
    table[some] = table[some] / 2.0
    assert device.as_device_alignment(table) is al1 or len(device._TABLE_CACHE) <= device._TABLE_CACHE_SLOTS
    # ... also when the edit keeps the keys AND the total mass (two values swapped / mass moved between two
    # patterns - a bootstrap refill of the same dict): the fingerprint covers every value, so this is a miss as well
    al_before = device.as_device_alignment(table)
    ks = list(table)
    k1 = ks[0]
    k2 = next(k for k in ks[1:] if table[k] != table[k1])
    before = sp.score_splits(table, splits[0:501:50])
    table[k1], table[k2] = table[k2], table[k1]
    al_swapped = device.as_device_alignment(table)
    assert al_swapped is not al_before
    swapped = dict(table)
    ref = np.array([O.dense_split_score(O.flattening(s_, swapped, "reduced")) for s_ in splits[0:501:250]])
    got = sp.score_splits(table, splits[0:501:50])
    assert np.max(np.abs(got[::5] - ref)) <= SCORE_TOL and not np.array_equal(got, before)


def test_config5_shape_12_taxa(sp):
    """BASELINE config 5 shape (simulated 12-taxon alignments x all 2035 splits): a 100 k-site table has 13.5 k patterns
    and 5|7, 6|6 flattenings of ~1800 x 1800 used ids - too much for one workgroup's LDS, so these splits run the same
    kernel with its arrays in global memory (HBM form, sparse.hip); the small table stays in LDS.  Both against the oracle."""
    from splitp_amd import simulation as sim
    from splitp_amd import synthetic as syn

    n = 12
    names = taxa_names(n)
    splits = list(sp.all_splits(names))
    assert len(splits) == 2035
    tree = syn.balanced_tree(n)
    for length, seed in ((20_000, 5), (100_000, 6)):
        dev = sim.generate_device_alignment(tree, sim.JukesCantor(), length, seed=seed, branch_length=0.05)
        dev.taxa = tuple(names)
        scores, status = sp.score_splits(dev, splits, return_status=True)
        assert np.all((status & 3) == 0) and np.all(np.isfinite(scores))
        keys, w, cnt = dev.fetch()
        true = syn.tree_splits(tree, n)
        is_true = np.array([frozenset(names.index(t) for t in a) in true or frozenset(names.index(t) for t in b) in true
                            for a, b in splits])
        assert is_true.sum() == 9 and scores[is_true].max() < scores[~is_true].min()
        for i in (0, 300, 1200, 2034):          # 2|10, 3|9, 5|7 and 6|6 splits
            oa = [names.index(t) for t in splits[i][0]]
            ob = [names.index(t) for t in splits[i][1]]
            M = O.reduced_flattening_packed(keys, cnt.astype(np.float64), n, oa, ob)[0]
            assert abs(O.dense_split_score(M) - scores[i]) <= SCORE_TOL, (length, i, M.shape)
        # the multi-alignment entry point + hand-back on the same table
        again = sp.score_splits(dev, splits[1900:2035])
        assert np.array_equal(again, scores[1900:2035])
        # the drop-in per-call form in the reference's default (sparse dok) format: a 4096 x 4096 matrix, beyond the
        # dense route - split_score re-expresses it as a table for the sparse kernel
        if length == 20_000:
            table = dict(dev.items())
            F = sp.flattening(splits[2034], table)
            assert F.shape == (4096, 4096)
            got = sp.split_score(F)
            assert isinstance(got, float) and abs(got - scores[2034]) <= 1e-12
        else:   # ... and the reduced (dense ndarray) format: 1812 x 1863 used rows / columns
            table = dict(dev.items())
            Fr = sp.flattening(splits[2034], table, sp.FlatFormat.reduced)
            assert min(Fr.shape) > 1024
            got = sp.split_score(Fr)
            assert isinstance(got, np.float64) and abs(got - scores[2034]) <= 1e-12


def test_sparse_kernel_edge_tables(sp):
    """Small and odd tables through the batched entry point (LDS form, its small-side path and the global-memory form):
    a single pattern, fewer than 5 patterns (min(shape) <= 4: exactly 0 like the reference), an all-constant alignment,
    3 and 4 taxa, and a 16-taxon table (largest n of the sparse kernel) against the oracle."""
    from splitp_amd import synthetic as syn

    # 4 taxa, the reference's own table: every flattening is 4 x 4 -> 0.0
    dev4 = sp.DeviceAlignment.from_table(REF4_TABLE, taxa=("0", "1", "2", "3"))
    s4 = sp.score_splits(dev4, [(("0", "1"), ("2", "3")), (("0", "2"), ("1", "3")), (("0",), ("1", "2", "3"))])
    assert np.array_equal(s4, np.zeros(3))
    # one pattern / all-constant alignment: rank 1 -> 0.0
    one = sp.DeviceAlignment.from_table({"ACGTAC": 1.0}, taxa=tuple("012345"))
    assert sp.score_splits(one, [(("0", "1", "2"), ("3", "4", "5"))])[0] == 0.0
    const = sp.DeviceAlignment.from_table({"AAAAAA": 0.25, "CCCCCC": 0.25, "GGGGGG": 0.25, "TTTTTT": 0.25}, taxa=tuple("012345"))
    assert sp.score_splits(const, [(("0", "1", "2"), ("3", "4", "5")), (("0", "1"), ("2", "3", "4", "5"))]).max() == 0.0
    # 6 taxa, short alignment: every split, LDS form, against the oracle (includes 5 x 5 .. 64 x 64 used shapes)
    names = taxa_names(6)
    keys, counts = syn.pattern_table(syn.simulate_sites(6, 3000, 0.3, seed=2))
    dev6 = sp.DeviceAlignment.from_arrays(keys, None, 6, counts=counts, n_sites=3000, taxa=names)
    splits6 = list(sp.all_splits(names))
    s6, st6 = sp.score_splits(dev6, splits6, return_status=True)
    assert np.all((st6 & 3) == 0)
    for i, spl in enumerate(splits6):
        M = O.reduced_flattening_packed(keys, counts.astype(np.float64), 6, [names.index(t) for t in spl[0]],
                                        [names.index(t) for t in spl[1]])[0]
        want = 0.0 if min(M.shape) <= 4 else O.dense_split_score(M)
        assert abs(want - s6[i]) <= SCORE_TOL, (i, M.shape)
    # 16 taxa (32-bit keys exactly full), 4000 sites: balanced 8|8 splits need the global-memory form
    g_names = taxa_names(16)
    keys, counts = syn.pattern_table(syn.simulate_sites(16, 4000, 0.05, seed=4))
    dev16 = sp.DeviceAlignment.from_arrays(keys, None, 16, counts=counts, n_sites=4000, taxa=g_names)
    rng = np.random.default_rng(0)
    some = []
    for k in (2, 5, 8):
        left = sorted(rng.choice(16, size=k, replace=False).tolist())
        some.append((tuple(g_names[t] for t in left), tuple(g_names[t] for t in range(16) if t not in left)))
    tree_split = (tuple(g_names[:8]), tuple(g_names[8:]))
    some.append(tree_split)
    s16, st16 = sp.score_splits(dev16, some, return_status=True)
    assert np.all((st16 & 3) == 0)
    for i, spl in enumerate(some):
        M = O.reduced_flattening_packed(keys, counts.astype(np.float64), 16, [g_names.index(t) for t in spl[0]],
                                        [g_names.index(t) for t in spl[1]])[0]
        assert abs(O.dense_split_score(M) - s16[i]) <= SCORE_TOL, (i, M.shape)


def test_sparse_kernel_random_tables(sp):
    """Randomised sweep: tables of 4..9 taxa, 30..4000 sites, short and long branches (few / many patterns, huge and unit
    counts), every split, batched sparse route against the oracle's dense SVD on the reduced flattening (exactly 0 where
    min(shape) <= 4, like the reference)."""
    from splitp_amd import synthetic as syn

    rng = np.random.default_rng(2024)
    worst = 0.0
    for trial in range(24):
        n = int(rng.integers(4, 10))
        if n % 2:
            n += 1 if n < 9 else -1
        length = int(rng.choice([30, 200, 1500, 4000]))
        branch = float(rng.choice([0.002, 0.05, 0.4, 2.0]))
        names = taxa_names(n)
        keys, counts = syn.pattern_table(syn.simulate_sites(n, length, branch, seed=100 + trial))
        if trial % 5 == 0:
            counts = counts * 70_000            # counts beyond 16 bits: entered as several table rows
        dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=int(counts.sum()), taxa=names)
        splits = list(sp.all_splits(names))
        got, st = sp.score_splits(dev, splits, return_status=True)
        assert np.all((st & 1) == 0)
        for i, spl in enumerate(splits):
            M = O.reduced_flattening_packed(keys, counts.astype(np.float64), n, [names.index(t) for t in spl[0]],
                                            [names.index(t) for t in spl[1]])[0]
            want = 0.0 if min(M.shape) <= 4 else O.dense_split_score(M)
            if np.isnan(want):          # the reference's unclamped dense path: operand rounded below zero
                want = 0.0
            err = abs(want - got[i]) if want > 1e-6 or got[i] > 1e-6 else abs(want ** 2 - got[i] ** 2)
            worst = max(worst, err)
            assert err <= SCORE_TOL, (trial, n, length, branch, i, M.shape, want, got[i])
    assert worst <= SCORE_TOL


def _copy_mutate_table(rng, n, length, letters):
    """Random alignment that is not tree-shaped like the simulator's: taxon t copies a random earlier taxon with its own
    per-site mutation probability; `letters` of the 4 states in use."""
    sites = np.empty((length, n), dtype=np.uint8)
    sites[:, 0] = rng.integers(0, letters, size=length)
    for t in range(1, n):
        src = sites[:, int(rng.integers(0, t))]
        p = float(rng.choice([0.001, 0.02, 0.15, 0.5]))
        mut = rng.random(length) < p
        sites[:, t] = np.where(mut, rng.integers(0, letters, size=length), src)
    keys = np.zeros(length, dtype=np.uint64)
    for t in range(n):
        keys = (keys << np.uint64(2)) | sites[:, t].astype(np.uint64)
    uk, cnt = np.unique(keys, return_counts=True)
    return uk, cnt.astype(np.int64)


def test_sparse_kernel_random_shapes(sp):
    """Second randomised sweep: 5..13 taxa (odd counts too), restricted alphabets (many unused ids on raw sides, ranks 1..4),
    random split sizes; LDS form, small-side path and global-memory form against the oracle."""
    rng = np.random.default_rng(77)
    checked = 0
    for trial in range(30):
        n = int(rng.integers(5, 14))
        length = int(rng.choice([20, 150, 900, 6000]))
        letters = int(rng.choice([2, 3, 4, 4]))
        keys, counts = _copy_mutate_table(rng, n, length, letters)
        names = taxa_names(n)
        dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=int(counts.sum()), taxa=names)
        if n <= 8:
            splits = list(sp.all_splits(names))
        else:
            splits = []
            for _ in range(24):
                k = int(rng.integers(2, n - 1))
                left = sorted(rng.choice(n, size=k, replace=False).tolist())
                splits.append((tuple(names[t] for t in left), tuple(names[t] for t in range(n) if t not in left)))
        got, st = sp.score_splits(dev, splits, return_status=True)
        assert np.all((st & 1) == 0)
        if n <= 10 and trial % 3 == 0:       # the dense route (int8-limb Gram) and the float-weight table (fp64 Gram) too
            def close(a, b):   # scores near 0 are square roots of rounding noise: compare their squares there
                small = (a < 1e-6) & (b < 1e-6)
                return np.where(small, np.abs(a * a - b * b), np.abs(a - b)).max() <= SCORE_TOL
            assert close(sp.score_splits(dev, splits, route="dense"), got)
            dev_w = sp.DeviceAlignment.from_arrays(keys, counts / float(counts.sum()), n, taxa=names, exact=False)
            assert close(sp.score_splits(dev_w, splits), got)
            mi = sp.score_splits(dev, splits[:6], method=sp.Method.mutual_information)
            for i in range(min(6, len(splits))):
                oa = [names.index(t) for t in splits[i][0]]
                ob = [names.index(t) for t in splits[i][1]]
                assert abs(O.rank1_divergence_packed(keys, counts / float(counts.sum()), n, oa, ob) - mi[i]) <= SCORE_TOL
        for i, spl in enumerate(splits):
            M = O.reduced_flattening_packed(keys, counts.astype(np.float64), n, [names.index(t) for t in spl[0]],
                                            [names.index(t) for t in spl[1]])[0]
            if min(M.shape) > 500:
                continue
            want = 0.0 if min(M.shape) <= 4 else O.dense_split_score(M)
            if np.isnan(want):
                want = 0.0
            err = abs(want - got[i]) if want > 1e-6 or got[i] > 1e-6 else abs(want ** 2 - got[i] ** 2)
            assert err <= SCORE_TOL, (trial, n, length, letters, len(keys), i, M.shape, want, got[i], hex(st[i]))
            checked += 1
    assert checked > 500


def test_subflattening_and_histogram_random(sp):
    """Randomised sweep of the other two routes: batched subflattening scores and matrices against the oracle (exact
    integer moments), and the device histogram of random sequences with invalid / lower-case characters against a NumPy
    count of the valid sites (reference splitp/parsers/fasta.py:48-63)."""
    rng = np.random.default_rng(31)
    for trial in range(10):
        n = int(rng.integers(4, 12))
        length = int(rng.choice([40, 700, 5000]))
        keys, counts = _copy_mutate_table(rng, n, length, int(rng.choice([3, 4])))
        names = taxa_names(n)
        total = int(counts.sum())
        dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=total, taxa=names)
        splits = []
        for _ in range(12):
            k = int(rng.integers(1, n))
            left = sorted(rng.choice(n, size=k, replace=False).tolist())
            splits.append((tuple(names[t] for t in left), tuple(names[t] for t in range(n) if t not in left)))
        got = sp.score_splits(dev, splits, method=sp.Method.subflattening)
        for i, spl in enumerate(splits):
            oa = [names.index(t) for t in spl[0]]
            ob = [names.index(t) for t in spl[1]]
            want_m = O.subflattening_packed(keys, counts / float(total), n, oa, ob)
            got_m = sp.subflattening(spl, dev)
            assert np.array_equal(np.rint(got_m * total), np.rint(want_m * total))
            want = 0.0 if min(want_m.shape) <= 4 else O.dense_split_score(want_m)
            if np.isnan(want):
                want = 0.0
            err = abs(want - got[i]) if want > 1e-6 or got[i] > 1e-6 else abs(want ** 2 - got[i] ** 2)
            assert err <= SCORE_TOL, (trial, n, length, i, want_m.shape, want, got[i])
    for trial in range(10):
        n = int(rng.integers(2, 15)) if trial < 6 else int(rng.choice([16, 17, 20, 24]))   # > 16 taxa: sort-based form only
        length = int(rng.choice([1, 63, 1000, 70_000]))
        alphabet = np.frombuffer(b"ACGTacgtN-?X", dtype=np.uint8)
        p = np.array([20, 20, 20, 20, 3, 3, 3, 3, 1, 1, 0.5, 0.5])
        seqs = alphabet[rng.choice(len(alphabet), size=(n, length), p=p / p.sum())]
        dev = sp.DeviceAlignment.from_sequences(seqs)
        code = np.full(256, 255, dtype=np.uint8)
        for ch, d in zip(b"ACGTacgt", (0, 1, 2, 3, 0, 1, 2, 3)):
            code[ch] = d
        digits = code[seqs]                                   # (n, L)
        valid = (digits != 255).all(axis=0)
        k64 = np.zeros(length, dtype=np.uint64)
        for t in range(n):
            k64 = (k64 << np.uint64(2)) | (digits[t] & 3).astype(np.uint64)
        uk, uc = np.unique(k64[valid], return_counts=True)
        keys, w, cnt = dev.fetch()
        assert np.array_equal(keys, uk) and np.array_equal(cnt, uc) and dev.info()["N"] == int(valid.sum())


def test_wide_block_fallback_12_taxa(sp):
    """12-taxon tables whose flattenings have a slowly decaying spectrum behind the 4th value: the 4-wide block runs out of
    half products, the dense route cannot take sides beyond 1024 rows, so the kernel's 8-wide fallback block has to finish
    them (api.hip run_sparse_route; status shows > 41 half products).  Found by the randomised hunt (tools/gpu_fuzz_long.py)."""
    went_wide = 0
    for seed in (1, 3):
        rng = np.random.default_rng(seed)
        n = 12
        keys, counts = _copy_mutate_table(rng, n, 20000, 3)
        names = taxa_names(n)
        dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=int(counts.sum()), taxa=names)
        splits = []
        for _ in range(16):
            k = int(rng.integers(2, n - 1))
            left = sorted(rng.choice(n, size=k, replace=False).tolist())
            splits.append((tuple(names[t] for t in left), tuple(names[t] for t in range(n) if t not in left)))
        got, st = sp.score_splits(dev, splits, return_status=True)
        assert not np.any(np.asarray(st) & 3), [hex(x) for x in st]
        went_wide += int(np.sum((np.asarray(st) >> 8) > 41))
        again = sp.score_splits(dev, splits)
        assert np.array_equal(got, again)                    # reproducible bit for bit, fallback included
        for i, spl in enumerate(splits):
            m = O.reduced_flattening_packed(keys, counts.astype(np.float64), n, [names.index(t) for t in spl[0]],
                                            [names.index(t) for t in spl[1]])[0]
            if min(m.shape) > 400:
                continue
            want = 0.0 if min(m.shape) <= 4 else O.dense_split_score(m)
            assert abs(want - got[i]) <= SCORE_TOL, (seed, i, m.shape, want, got[i], hex(st[i]))
    assert went_wide > 0   # (if this fires the tables no longer reach the fallback: pick harder ones)


def test_flattening_api_random_tables(sp):
    """flattening() on random small tables in the reference's own input form (dict of pattern string -> probability),
    both formats, every split form, bit-exact against the loops-level port of constructions.py:31-102."""
    rng = np.random.default_rng(77)
    for trial in range(30):
        n = int(rng.integers(2, 8))
        length = int(rng.choice([7, 50, 400]))
        letters = int(rng.choice([2, 4, 4]))
        keys, counts = _copy_mutate_table(rng, n, length, letters)
        total = int(counts.sum())
        table = {}
        for k, c in zip(keys.tolist(), counts.tolist()):
            table["".join("ACGT"[(k >> (2 * (n - 1 - t))) & 3] for t in range(n))] = c / total
        names = [str(t) for t in range(n)]
        perm = rng.permutation(n)
        a = int(rng.integers(1, n))
        left, right = sorted(perm[:a].tolist()), sorted(perm[a:].tolist())
        forms = [(tuple(str(t) for t in left), tuple(str(t) for t in right)),
                 ({str(t) for t in left}, {str(t) for t in right})]
        if n <= 10:
            forms.append("".join(map(str, left)) + "|" + "".join(map(str, right)))
        for split in forms:
            want_r = O.flattening(split, table, "reduced")
            got_r = sp.flattening(split, table, sp.FlatFormat.reduced)
            assert got_r.shape == want_r.shape and np.array_equal(got_r, want_r), (trial, split)
            want_s = O.flattening(split, table, "sparse")
            got_s = sp.flattening(split, table, sp.FlatFormat.sparse)
            assert got_s.shape == want_s.shape and (abs(got_s.tocoo() - want_s.tocoo())).nnz == 0, (trial, split)
            w = O.split_score(want_r)
            g = sp.split_score(got_r)
            if np.isnan(w):
                assert np.isnan(g) or g * g <= 1e-13
            else:
                assert abs(w - g) <= SCORE_TOL or abs(w * w - g * g) <= 5e-14, (trial, split, w, g)
        assert names


def test_config4_size_subflattening(sp):
    """BASELINE config 4 size (20 taxa, 1M bp: 40-bit pattern keys, 61 x 61 moment matrix, blocks up to 31 x 31):
    exact moment identity and oracle spot checks on splits of every size class."""
    from splitp_amd import synthetic as syn

    n, length = 20, 1_000_000
    sites = syn.simulate_sites(n, length, 0.05, seed=4)
    keys, counts = syn.pattern_table(sites)
    assert int(keys.max()) >= 1 << 32                       # the keys really need more than 32 bits
    names = syn.taxa_names(n)
    dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=length, taxa=names)
    rng = np.random.default_rng(4)
    splits = []
    for k in range(2, 11):
        for _ in range(40):
            left = sorted(rng.choice(n, size=k, replace=False).tolist())
            splits.append((tuple(names[t] for t in left), tuple(names[t] for t in range(n) if t not in left)))
    got, st = sp.score_splits(dev, splits, method=sp.Method.subflattening, return_status=True)
    assert not np.any(st & 3)
    # the whole candidate set through the NumPy enumeration: 524 267 splits, one call
    every = sp.score_all_splits(dev, method=sp.Method.subflattening)
    assert every.shape == (524267,) and np.all(np.isfinite(every)) and every.min() >= 0 and every.max() < 1
    from itertools import islice
    head = list(islice(sp.all_splits(names), 64))
    assert np.array_equal(sp.score_splits(dev, head, method=sp.Method.subflattening), every[:64])
    # ... which enumerates the splits on the device: same scores as the host enumeration, every size class and option
    from splitp_amd import batch, _lib
    ta, aa = batch.encode_all_splits(n)
    host_way, _ = batch.score_encoded(dev, ta, aa, _lib.SP_METHOD_SUBFLATTENING)
    assert np.array_equal(host_way, every)
    for kw in ({"trivial": True}, {"size": 10}, {"size": 1}, {"size": 7}):
        ta, aa = batch.encode_all_splits(n, **kw)
        want, _ = batch.score_encoded(dev, ta, aa, _lib.SP_METHOD_SUBFLATTENING)
        assert np.array_equal(sp.score_all_splits(dev, method=sp.Method.subflattening, **kw), want), kw
    M = O.moment_matrix(keys, counts, n)
    for i in range(0, len(splits), 17):
        oa = [names.index(t) for t in splits[i][0]]
        ob = [names.index(t) for t in splits[i][1]]
        S = M[np.ix_(O.subflattening_index(oa, n), O.subflattening_index(ob, n))] / float(length)
        m_gpu = sp.subflattening(splits[i], dev)
        assert np.array_equal(np.rint(m_gpu * length), np.rint(S * length))
        assert abs(O.dense_split_score(S) - got[i]) <= SCORE_TOL
    # the full enumeration (`every`, all_splits order) against the oracle on at least four splits of EVERY size class -
    # first, last and two inside - like config 3 (every 17th of 360 random splits was 22 checks of 524 267)
    ta, aa = batch.encode_all_splits(n)
    checked = 0
    for k in range(2, 11):
        idx = np.nonzero(np.minimum(aa, n - aa) == k)[0]
        for i in sorted({int(idx[0]), int(idx[len(idx) // 3]), int(idx[2 * len(idx) // 3]), int(idx[-1])}):
            oa, ob = ta[i][:aa[i]].tolist(), ta[i][aa[i]:].tolist()
            S = M[np.ix_(O.subflattening_index(oa, n), O.subflattening_index(ob, n))] / float(length)
            assert abs(O.dense_split_score(S) - every[i]) <= SCORE_TOL, (k, i)
            checked += 1
    assert checked >= 36


@pytest.mark.gpu
def test_subflattening_score_workgroup_shapes_agree(sp):
    """The fast subflattening score kernel picks its workgroup shape per launch (waves per workgroup x workgroups per CU,
    the per-wave LDS area sized for the longest side of the batch): every shape the option `subscore_waves` can pin gives
    bit-identical scores - a wave's arithmetic does not depend on where it runs - for a batch of every size class and for
    one of short sides only (another LDS pitch)."""
    from splitp_amd import synthetic as syn

    n = 14
    sites = syn.simulate_sites(n, 100_000, 0.05, seed=19)
    keys, counts = syn.pattern_table(sites)
    names = syn.taxa_names(n)
    dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=100_000, taxa=names)
    allsp = list(sp.all_splits(names))
    for splits in (allsp[::5], [s for s in allsp if min(len(s[0]), len(s[1])) <= 3][::3]):
        ref = sp.score_splits(dev, splits, method=sp.Method.subflattening)
        assert np.all(np.isfinite(ref))
        try:
            for wv in (1, 3, 4, 7, 8, 12, 16):
                sp.get_context().set_option("subscore_waves", wv)
                got = sp.score_splits(dev, splits, method=sp.Method.subflattening)
                assert np.array_equal(got, ref), wv
        finally:
            sp.get_context().set_option("subscore_waves", 0)


def test_subflattening_score_kernels_agree(sp, monkeypatch):
    """The two eigen-solvers behind the batched subflattening score - Householder tridiagonalisation + Sturm multisection
    (default up to 20 taxa) and the cyclic Jacobi kernel (larger tables; forced here by SPLITP_SUBSCORE_JACOBI) - on the same
    Gram matrices: all size classes of a 14-taxon table, a float-weight table, and a degenerate one."""
    from splitp_amd import synthetic as syn

    n = 14
    sites = syn.simulate_sites(n, 200_000, 0.05, seed=9)
    keys, counts = syn.pattern_table(sites)
    names = syn.taxa_names(n)
    splits = list(sp.all_splits(names))[::7]
    tables = [sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=200_000, taxa=names),
              sp.DeviceAlignment.from_arrays(keys, counts / 200_000.0, n, taxa=names, exact=False),
              sp.DeviceAlignment.from_arrays(keys[:3], None, n, counts=counts[:3], n_sites=int(counts[:3].sum()), taxa=names)]
    for dev in tables:
        sp.get_context().set_option("subscore_jacobi", 0)
        fast, st = sp.score_splits(dev, splits, method=sp.Method.subflattening, return_status=True)
        assert not np.any(st & 3)
        sp.get_context().set_option("subscore_jacobi", 1)
        try:
            slow = sp.score_splits(dev, splits, method=sp.Method.subflattening)
        finally:
            sp.get_context().set_option("subscore_jacobi", 0)
        both_nan = np.isnan(fast) & np.isnan(slow)
        err = np.where(both_nan, 0.0, np.abs(fast - slow))
        err2 = np.where(both_nan, 0.0, np.abs(fast ** 2 - slow ** 2))
        assert np.all((err <= 1e-11) | (err2 <= 1e-13)), (float(np.nanmax(err)), float(np.nanmax(err2)))


def test_moment_matrix_kernels_agree(sp):
    """The signed second-moment matrix behind every subflattening by its two kernels - S^T diag(w) S on the
    fp64 matrix cores (default up to 21 taxa) and round 1's vector kernels with int64 partial sums (option moments_valu) -
    against the oracle: count tables EXACT (integers), float-weight tables to 1e-13, 4 - 21 taxa, run-to-run identical."""
    import ctypes as C

    from splitp_amd import _lib
    from splitp_amd import synthetic as syn

    ctx = sp.get_context()
    lib = ctx._lib

    def moments(keys, counts, n, length, exact, valu):
        ctx.set_option("moments_valu", valu)
        names = syn.taxa_names(n)
        dev = (sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=length, taxa=names) if exact else
               sp.DeviceAlignment.from_arrays(keys, counts / float(length), n, taxa=names, exact=False))
        m = 3 * n + 1
        out_i, out_f = np.zeros(m * m, dtype=np.int64), np.zeros(m * m, dtype=np.float64)
        _lib.check(lib.sp_moment_matrix(dev.handle, _lib._ptr(out_i, C.c_int64) if exact else None,
                                        None if exact else _lib._ptr(out_f, C.c_double)))
        ctx.set_option("moments_valu", 0)
        return (out_i if exact else out_f).reshape(m, m)

    for n, length, seed in ((4, 300, 1), (7, 5_000, 2), (12, 100_000, 3), (17, 200_000, 4), (20, 300_000, 5), (21, 20_000, 6)):
        sites = syn.simulate_sites(n + (n & 1), length, 0.05, seed=seed)[:, :n]
        keys, counts = syn.pattern_table(sites)
        want = O.moment_matrix(keys, counts, n)
        new = moments(keys, counts, n, length, True, 0)
        assert np.array_equal(new, want), n
        assert np.array_equal(moments(keys, counts, n, length, True, 1), want), n
        assert np.array_equal(moments(keys, counts, n, length, True, 0), new), n
        # counts near 2^32 (one pattern carries most sites): products and partial sums stay below 2^53
        big = counts.astype(np.int64).copy()
        big[0] = 4_000_000_000
        assert np.array_equal(moments(keys, big, n, int(big.sum()), True, 0), O.moment_matrix(keys, big, n)), n
        ref = want / float(length)
        newf = moments(keys, counts, n, length, False, 0)
        assert np.abs(newf - ref).max() <= 1e-13 and np.abs(moments(keys, counts, n, length, False, 1) - ref).max() <= 1e-13
        assert np.array_equal(moments(keys, counts, n, length, False, 0), newf)


def test_subflattening_pair_and_single_kernels_agree(sp):
    """The default subflattening score kernel takes two splits of one size class per wave (csrc/subflat_pair.hip); option
    `subscore_pair` = 0 selects round 3's one-split-a-wave kernel.  Same Gram matrices, two implementations of Householder +
    Sturm: scores within 1e-11 (or 1e-13 in score^2) of each other for every split of 6 - 14 taxon tables (count, float-weight,
    degenerate; with and without the trivial splits: classes of 4 rows, odd and even class sizes), none flagged.  A split's
    score must not depend on its partner in the wave: a shuffled list (other pairs) returns the bits of the enumeration."""
    from splitp_amd import synthetic as syn

    ctx = sp.get_context()
    for n, length, seed in ((6, 5000, 1), (9, 30_000, 2), (12, 80_000, 3), (14, 100_000, 4)):
        sites = syn.simulate_sites(n + (n & 1), length, 0.05, seed=seed)[:, :n]   # (balanced trees need an even leaf count)
        keys, counts = syn.pattern_table(sites)
        names = syn.taxa_names(n)
        tables = [sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=length, taxa=names),
                  sp.DeviceAlignment.from_arrays(keys, counts / float(length), n, taxa=names, exact=False),
                  sp.DeviceAlignment.from_arrays(keys[:3], None, n, counts=counts[:3], n_sites=int(counts[:3].sum()), taxa=names)]
        for ti, dev in enumerate(tables):
            for trivial in (False, True):
                res = {}
                for pair in (1, 0):
                    ctx.set_option("subscore_pair", pair)
                    res[pair], st = sp.score_all_splits(dev, method=sp.Method.subflattening, trivial=trivial, return_status=True)
                    assert not np.any(st & 3), (n, ti, pair)
                ctx.set_option("subscore_pair", 1)
                a, b = res[1], res[0]
                assert np.array_equal(np.isnan(a), np.isnan(b))
                nan = np.isnan(a)
                d = np.where(nan, 0.0, np.abs(a - b))
                d2 = np.where(nan, 0.0, np.abs(a * a - b * b))
                assert np.all((d <= 1e-11) | (d2 <= 1e-13)), (n, ti, trivial, float(d.max()), float(d2.max()))
        dev = tables[0]
        allsp = list(sp.all_splits(names))
        full = dict(zip(allsp, sp.score_all_splits(dev, method=sp.Method.subflattening)))
        rng = np.random.default_rng(seed)
        pick = [allsp[i] for i in rng.permutation(len(allsp))[:777]]
        got = sp.score_splits(dev, pick, method=sp.Method.subflattening)
        assert np.array_equal(got, np.array([full[s] for s in pick])), n
        # one split alone (its own partner) and a list of one class with an odd count
        assert sp.score_splits(dev, pick[:1], method=sp.Method.subflattening)[0] == full[pick[0]]
        odd = [s for s in allsp if min(len(s[0]), len(s[1])) == n // 2][:5]
        assert np.array_equal(sp.score_splits(dev, odd, method=sp.Method.subflattening), np.array([full[s] for s in odd]))


def test_device_simulator_20_taxa(sp):
    """The device simulator beyond 16 taxa (40-bit site words, sort-based histogram): reproducible, every site counted,
    and the pairwise mismatch fractions of the 20-taxon balanced tree match Jukes-Cantor along the path between the
    leaves (1 / 2 / 4 branches of 0.05 for taxa 0-1, 0-2, 0-3: 3/4 (1 - exp(-4 d / 3)))."""
    from splitp_amd import simulation as sim
    from splitp_amd import synthetic as syn

    n, length = 20, 400_000
    tree = syn.balanced_tree(n)
    dev = sim.generate_device_alignment(tree, sim.JukesCantor(), length, seed=11, branch_length=0.05)
    keys, w, cnt = dev.fetch()
    assert dev.info()["N"] == length and int(cnt.sum()) == length and np.all(np.diff(keys.astype(np.int64)) > 0)
    again = sim.generate_device_alignment(tree, sim.JukesCantor(), length, seed=11, branch_length=0.05).fetch()
    assert np.array_equal(again[0], keys) and np.array_equal(again[2], cnt)
    digit = lambda t: (keys >> np.uint64(2 * (n - 1 - t))) & np.uint64(3)
    # path lengths in the balanced tree ((((0,1),2),(3,4)),...): derive them from the nested tuples
    def depth_paths(node, path, out):
        if isinstance(node, tuple):
            for i, ch in enumerate(node):
                depth_paths(ch, path + [(id(node), i)], out)
        else:
            out[node] = path
    paths = {}
    depth_paths(tree, [], paths)
    for other in (1, 2, 3, 10, 19):
        pa, pb = paths[0], paths[other]
        common = 0
        while common < min(len(pa), len(pb)) and pa[common] == pb[common]:
            common += 1
        # the root has no branch of its own: its two children hang on it directly
        d = 0.05 * ((len(pa) - common) + (len(pb) - common))
        p = 0.75 * (1.0 - np.exp(-4.0 * d / 3.0))
        frac = float(cnt[digit(0) != digit(other)].sum()) / length
        assert abs(frac - p) <= 5.0 * np.sqrt(p * (1 - p) / length), (other, d, frac, p)


def test_big_table_form(sp, golden, monkeypatch):
    """The big-table form of the sparse route (k_sparse_big: stable segmented sorts + chunked segmented sums, everything in
    global memory) - the only route for 12+ taxa once a table has more than 65535 patterns or float weights.
    (a) forced on the 10-taxon golden table: the reference's scores, counts and float weights;
    (b) a 12-taxon 1 M-site table (124 k patterns): oracle on the short-side splits, repeatable, floats = counts;
    (c) forced on an adversarial 12-taxon table that needs its 8-wide fallback block: agrees with the list kernels."""
    from splitp_amd import synthetic as syn

    g, names, splits, table = _n10(golden, "n10_L100k")
    dev = sp.DeviceAlignment.from_table(table, taxa=names)
    dev_w = sp.DeviceAlignment.from_arrays(g["keys"], g["probs"], 10, taxa=names, exact=False)
    sp.get_context().set_option("force_big", 1)
    for d in (dev, dev_w):
        got, st = sp.score_splits(d, splits, return_status=True)
        assert not np.any(st & 3) and np.abs(got - g["scores"]).max() <= SCORE_TOL
        sp.get_context().set_option("big_by_keys", 1)          # compaction by sorting the raw side keys (sides > 14 taxa)
        assert np.array_equal(sp.score_splits(d, splits), got)
        sp.get_context().set_option("big_by_keys", 0)
    sp.get_context().set_option("force_big", 0)

    n, length = 12, 1_000_000
    names12 = syn.taxa_names(n)
    sites = syn.simulate_sites(n, length, 0.08, seed=2)
    keys, counts = syn.pattern_table(sites)
    assert len(keys) > 65535
    big = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=length, taxa=names12)
    allsp = list(sp.all_splits(names12))
    pick = allsp[::29]
    got, st = sp.score_splits(big, pick, return_status=True)
    assert not np.any(st & 3)
    assert np.array_equal(got, sp.score_splits(big, pick))
    big_w = sp.DeviceAlignment.from_arrays(keys, counts / float(length), n, taxa=names12, exact=False)
    assert np.abs(sp.score_splits(big_w, pick) - got).max() <= SCORE_TOL
    checked = 0
    for i, spl in enumerate(pick):
        if min(len(spl[0]), len(spl[1])) > 3:
            continue
        m = O.reduced_flattening_packed(keys, counts.astype(np.float64), n, [names12.index(t) for t in spl[0]],
                                        [names12.index(t) for t in spl[1]])[0]
        assert abs(O.dense_split_score(m) - got[i]) <= SCORE_TOL, (i, m.shape)
        checked += 1
    assert checked >= 5

    rng = np.random.default_rng(1)
    keys2, counts2 = _copy_mutate_table(rng, 12, 20000, 3)
    adv = sp.DeviceAlignment.from_arrays(keys2, None, 12, counts=counts2, n_sites=int(counts2.sum()), taxa=names12)
    splits2 = []
    for _ in range(16):
        k = int(rng.integers(2, 11))
        left = sorted(rng.choice(12, size=k, replace=False).tolist())
        splits2.append((tuple(names12[t] for t in left), tuple(names12[t] for t in range(12) if t not in left)))
    want = sp.score_splits(adv, splits2)
    sp.get_context().set_option("force_big", 1)
    got2, st2 = sp.score_splits(adv, splits2, return_status=True)
    sp.get_context().set_option("force_big", 0)
    assert not np.any(st2 & 3) and np.abs(got2 - want).max() <= SCORE_TOL
    assert int((st2 >> 8).max()) > 41       # at least one split went through the wide block


def test_flattening_scores_20_taxa(sp):
    """Flattening scores beyond 16 taxa (big-table form; sides up to 14 taxa): a 20-taxon 100 k-site alignment against the
    reference's own sparse scorer (ARPACK top-4 + Frobenius norm, phylogenetics.py:303-312) on the reduced flattening."""
    import scipy.sparse
    from splitp_amd import synthetic as syn

    n, length = 20, 100_000
    names = syn.taxa_names(n)
    sites = syn.simulate_sites(n, length, 0.05, seed=6)
    keys, counts = syn.pattern_table(sites)
    dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=length, taxa=names)
    rng = np.random.default_rng(6)
    splits = [(tuple(names[:10]), tuple(names[10:]))]                      # the root split of the balanced tree
    for k in (6, 8, 10, 13):
        left = sorted(rng.choice(n, size=k, replace=False).tolist())
        splits.append((tuple(names[t] for t in left), tuple(names[t] for t in range(n) if t not in left)))
    got, st = sp.score_splits(dev, splits, return_status=True)
    assert not np.any(st & 3) and np.array_equal(got, sp.score_splits(dev, splits))
    for i, spl in enumerate(splits):
        rows, cols = O.flat_indices(keys, n, [names.index(t) for t in spl[0]], [names.index(t) for t in spl[1]])
        ur, ri = np.unique(rows, return_inverse=True)
        uc, ci = np.unique(cols, return_inverse=True)
        m = scipy.sparse.coo_matrix((counts.astype(np.float64), (ri, ci)), shape=(len(ur), len(uc))).tocsc()
        want = O.sparse_split_score(m)
        assert abs(want - got[i]) <= 1e-9, (i, m.shape, want, got[i])
    assert got[0] < got[1:].min()                                           # the true split scores lowest
    # sides of more than 14 taxa are beyond the bitmap compaction: the raw side keys are sorted instead
    long_side = [(tuple(names[:2]), tuple(names[2:])), ((names[0], names[7], names[19]), tuple(t for t in names if t not in (names[0], names[7], names[19])))]
    got_l, st_l = sp.score_splits(dev, long_side + splits[:2], return_status=True)
    assert not np.any(st_l & 3)
    assert np.abs(got_l[2:] - got[:2]).max() <= SCORE_TOL                   # same answers as the bitmap path for the short sides
    for i, spl in enumerate(long_side):
        rows, cols = O.flat_indices(keys, n, [names.index(t) for t in spl[0]], [names.index(t) for t in spl[1]])
        ur, ri = np.unique(rows, return_inverse=True)
        uc, ci = np.unique(cols, return_inverse=True)
        m = np.zeros((len(ur), len(uc)))
        np.add.at(m, (ri, ci), counts.astype(np.float64))
        assert abs(O.dense_split_score(m) - got_l[i]) <= SCORE_TOL, (i, m.shape)


def test_score_all_splits_matches_host_enumeration(sp):
    """sp_score_all_splits (splits un-ranked on the device) against the host enumeration of all_splits, every taxon count
    from 2 to 13 and every option; count-only call; methods other than subflattening are refused."""
    from splitp_amd import batch, _lib
    import ctypes as C

    rng = np.random.default_rng(21)
    for n in range(2, 14):
        keys, counts = _copy_mutate_table(rng, n, 3000, 4)
        names = taxa_names(n)
        dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=int(counts.sum()), taxa=names)
        for kw in ({}, {"trivial": True}, {"size": max(1, n // 2)}, {"size": 1}):
            ta, aa = batch.encode_all_splits(n, **kw)
            got, st = sp.score_all_splits(dev, method=sp.Method.subflattening, return_status=True, **kw)
            assert len(got) == len(aa) == len(list(sp.all_splits(names, **kw))), (n, kw)
            if len(aa):
                want, _ = batch.score_encoded(dev, ta, aa, _lib.SP_METHOD_SUBFLATTENING)
                assert np.array_equal(got, want, equal_nan=True) and not np.any(st & 3), (n, kw)
        cnt = C.c_int64()
        _lib.check(dev.ctx._lib.sp_score_all_splits(dev.handle, _lib.SP_METHOD_SUBFLATTENING, 0, 0, C.byref(cnt), None, None, None))
        assert cnt.value == len(list(sp.all_splits(names)))
    # (the flattening routes enumerate on the device too since round 2: test_score_all_splits_flattening_planned_on_device)
    _lib.check(dev.ctx._lib.sp_score_all_splits(dev.handle, _lib.SP_METHOD_FLATTENING, 0, 0, C.byref(cnt), None, None, None))
    assert cnt.value == len(list(sp.all_splits(names)))
    with pytest.raises(ValueError):
        _lib.check(dev.ctx._lib.sp_score_all_splits(dev.handle, 99, 0, 0, C.byref(cnt), None, None, None))


def test_wide_block_plateau_regression(sp, monkeypatch):
    """A table the randomised hunt found (seed 7000, trial 101): three large singular values, then a pair 4e-5 apart and a
    4-fold degenerate group.  The 4-wide phase stalls on the 5th vector of the pair; in the 8-wide block the top-4 Ritz sum
    then sits on a plateau until the direction of the 4th value has grown out of the guard columns and overtakes it.
    Every route has to return the reference's score."""
    import os
    from tests.conftest import GOLDEN

    d = np.load(os.path.join(GOLDEN, "regress_wide_plateau.npz"))
    n = int(d["n"])
    names = taxa_names(n)
    keys, counts = d["keys"], d["counts"]
    spl = [(tuple(names[t] for t in d["oa"]), tuple(names[t] for t in d["ob"]))]
    m = O.reduced_flattening_packed(keys, counts.astype(np.float64), n, list(d["oa"]), list(d["ob"]))[0]
    want = O.dense_split_score(m)
    dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=int(counts.sum()), taxa=names)
    dev_w = sp.DeviceAlignment.from_arrays(keys, counts / float(counts.sum()), n, taxa=names, exact=False)
    assert abs(sp.score_splits(dev, spl)[0] - want) <= SCORE_TOL
    assert abs(sp.score_splits(dev, spl, route="dense")[0] - want) <= SCORE_TOL
    assert abs(sp.score_splits(dev_w, spl)[0] - want) <= SCORE_TOL
    sp.get_context().set_option("force_big", 1)
    got, st = sp.score_splits(dev, spl, return_status=True)
    assert abs(got[0] - want) <= SCORE_TOL and not (st[0] & 3)
    assert abs(sp.score_splits(dev_w, spl)[0] - want) <= SCORE_TOL


def test_plan_abi_errors_and_refcount(sp, golden):
    """sp_plan / sp_score_plan_async misuse is answered with status codes, never a crash: NULL arguments, a split that
    does not cover the taxa, an alignment of another taxon count, a float-weight table (no sparse route), and the
    reference count keeps a plan alive until its last holder lets go."""
    import ctypes as C
    import torch
    from splitp_amd import _lib, batch

    g = golden("n10_L10k")
    names = taxa_names(10)
    dev = sp.DeviceAlignment.from_table(O.unpack_table(g["keys"], g["probs"], 10), taxa=names)
    lib, ctx = dev.ctx._lib, dev.ctx
    taxa_arr, a_arr = sp.encode_all_splits(10)
    h = C.c_void_p()
    assert lib.sp_plan_create(None, 10, _lib._ptr(taxa_arr, C.c_int32), _lib._ptr(a_arr, C.c_int32), 501, C.byref(h)) == _lib.SP_EINVAL
    bad = taxa_arr.copy()
    bad[3, 0] = bad[3, 1]                                           # a taxon twice
    assert lib.sp_plan_create(ctx.handle, 10, _lib._ptr(bad, C.c_int32), _lib._ptr(a_arr, C.c_int32), 501, C.byref(h)) == _lib.SP_EINVAL
    assert b"twice" in lib.sp_last_error()
    plan = batch.SplitPlan(ctx, taxa_arr, a_arr, 10)
    n, s = C.c_int(), C.c_int64()
    assert lib.sp_plan_info(plan.handle, C.byref(n), C.byref(s)) == 0 and (n.value, s.value) == (10, 501)
    sc = torch.zeros(501, dtype=torch.float64, device="cuda")
    st = torch.zeros(501, dtype=torch.int32, device="cuda")
    one = (C.c_void_p * 1)(dev.handle.value)
    assert lib.sp_score_plan_async(ctx.handle, one, 1, plan.handle, None, C.c_void_p(st.data_ptr())) == _lib.SP_EINVAL
    keys8, counts8 = np.arange(50, dtype=np.uint64), np.ones(50, dtype=np.int64)
    dev8 = sp.DeviceAlignment.from_arrays(keys8, None, 8, counts=counts8, n_sites=50, taxa=taxa_names(8))
    other = (C.c_void_p * 1)(dev8.handle.value)
    assert lib.sp_score_plan_async(ctx.handle, other, 1, plan.handle, C.c_void_p(sc.data_ptr()), C.c_void_p(st.data_ptr())) == _lib.SP_EINVAL
    dev_w = sp.DeviceAlignment.from_arrays(g["keys"], g["probs"], 10, taxa=names, exact=False)
    fl = (C.c_void_p * 1)(dev_w.handle.value)
    assert lib.sp_score_plan_async(ctx.handle, fl, 1, plan.handle, C.c_void_p(sc.data_ptr()), C.c_void_p(st.data_ptr())) == _lib.SP_ELIMIT
    assert b"float weights" in lib.sp_last_error()
    # retain / release: the Python object's finalizer drops the creator's reference, ours keeps the plan usable
    assert lib.sp_plan_retain(plan.handle) == 0
    handle = C.c_void_p(plan.handle.value)
    del plan
    import gc
    gc.collect()
    assert lib.sp_score_plan_async(ctx.handle, one, 1, handle, C.c_void_p(sc.data_ptr()), C.c_void_p(st.data_ptr())) == 0
    torch.cuda.synchronize()
    assert np.abs(sc.cpu().numpy() - g["scores"]).max() <= SCORE_TOL and not np.any(st.cpu().numpy() & 3)
    assert lib.sp_plan_release(handle) == 0


def test_bench_line_contract(sp):
    """`python bench.py` (short) on this GPU: ONE JSON line on stdout with the contract's keys, the roofline block
    (bound / achieved / peak / unit / frac / traffic + the binding counters) and a consistent value."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "60", "--warmup", "6", "--no-cpu-baseline",
                        "--spinup", "0.2"], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 60 and d["dtype"] == "f64" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and d["config"]["unconverged_splits_in_timed_region"] == 0
    assert abs(d["value"] - 501 * 60 / (d["ms_per_step"] * 60 * 1e-3)) <= 1e-6 * d["value"]
    assert d["host_us_per_step"] > 0 and d["config"]["steps_per_host_call"] >= 1
    roof = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "survey_8d", "pmc_record"):
        assert key in roof, key
    assert roof["bound"] == "lds"
    if roof["pmc_record"]:
        # a counter record whose source hash and workload shape match this tree: work- and busy-based fractions
        assert 0.05 < roof["frac"] < 1.0 and roof["traffic"] > 0
        assert 0.1 < roof["binding"]["bank_conflict_share"] < 0.8 and roof["binding"]["real_work"]["fma"] > 1e7
        useful = roof["binding"]["useful"]
        assert 0 < useful["conflict_free_lds_frac"] < roof["frac"] and 0 < useful["fp64_frac_of_peak"] < 0.2
        assert 0 < useful["lds_gather_frac_of_read_peak"] < 0.5
    else:
        # no record of THIS kernel build: the fraction is withheld, with the reason (VERDICT r2 / ADVICE r2: a constant
        # read from a file must not describe another kernel)
        assert roof["frac"] is None and roof["achieved"] is None and roof["pmc_record_reason"]
    # BASELINE's "dense flattening" pipeline, timed per phase in the same (default) invocation
    ns = d["north_star_pipeline"]
    assert ns["steps"] == 30 and ns["ms_per_step"] > 0 and ns["max_abs_score_diff_vs_sparse_route"] <= 1e-10
    assert all(ns["phase_ms_per_step"][k] > 0 for k in ("reindex", "scatter", "gram", "eigen"))
    assert abs(sum(ns["phase_ms_per_step"][k] for k in ("reindex", "scatter", "gram", "eigen")) - ns["ms_per_step"]) < 0.5 * ns["ms_per_step"]


def test_score_all_splits_flattening_planned_on_device(sp, golden):
    """score_all_splits on the flattening routes: the splits are enumerated AND (default route) planned on the device -
    same scores, bit for bit, as the host-planned call on list(all_splits(taxa)); trivial class, one size class, the
    dense route, the mutual-information score and a float-weight table (host-planned fallbacks) included."""
    g = golden("n10_L100k")
    names = taxa_names(10)
    dev = sp.DeviceAlignment.from_table(O.unpack_table(g["keys"], g["probs"], 10), taxa=names)
    splits = list(sp.all_splits(names))
    got, st = sp.score_all_splits(dev, return_status=True)
    assert np.array_equal(got, sp.score_splits(dev, splits)) and not np.any(st & 3)
    assert np.abs(got - g["scores"]).max() <= SCORE_TOL
    triv = list(sp.all_splits(names, trivial=True))
    assert np.array_equal(sp.score_all_splits(dev, trivial=True), sp.score_splits(dev, triv))
    four = list(sp.all_splits(names, size=4))
    assert np.array_equal(sp.score_all_splits(dev, size=4), sp.score_splits(dev, four)) and len(four) == 210
    assert np.array_equal(sp.score_all_splits(dev, route="dense"), sp.score_splits(dev, splits, route="dense"))
    assert np.array_equal(sp.score_all_splits(dev, method=sp.Method.mutual_information),
                          sp.score_splits(dev, splits, method=sp.Method.mutual_information))
    dev_w = sp.DeviceAlignment.from_arrays(g["keys"], g["probs"], 10, taxa=names, exact=False)
    assert np.abs(sp.score_all_splits(dev_w) - g["scores"]).max() <= SCORE_TOL
    # 12 taxa (most splits beyond the LDS form: the device chain) and a forced hand-back to the dense route
    from splitp_amd import simulation as sim
    from splitp_amd import synthetic as syn
    d12 = sim.generate_device_alignment(syn.balanced_tree(12), sim.JukesCantor(), 50_000, seed=4, branch_length=0.05)
    d12.taxa = tuple(taxa_names(12))
    s12 = sp.score_all_splits(d12)
    assert len(s12) == 2035 and np.array_equal(s12, sp.score_splits(d12, list(sp.all_splits(taxa_names(12)))))
    rng = np.random.default_rng(3)
    rk = np.unique(rng.integers(0, 4 ** 10, size=3000).astype(np.uint64))
    rc = rng.integers(1, 40, size=len(rk)).astype(np.int64)
    flat = sp.DeviceAlignment.from_arrays(rk, None, 10, counts=rc, n_sites=int(rc.sum()), taxa=names)
    fa, fst = sp.score_all_splits(flat, size=3, return_status=True)
    fb = sp.score_splits(flat, list(sp.all_splits(names, size=3)))
    assert np.array_equal(fa, fb) and not np.any(fst & 2)


def test_score_all_splits_shards(sp, golden):
    """sp_score_all_splits_shard: rank r of P enumerates the combinations r, r + P, ... of every size class on the device;
    the shards of P = 1, 2, 3 and 8 ranks, un-permuted with batch.shard_layout, reproduce the full call bit for bit -
    subflattening (16 taxa) and the device-planned flattening route (10 taxa), scores and status into device buffers."""
    import torch
    from splitp_amd import batch, _lib

    g = golden("n10_L100k")
    names = taxa_names(10)
    dev = sp.DeviceAlignment.from_table(O.unpack_table(g["keys"], g["probs"], 10), taxa=names)
    g16 = golden("n16_L4k")
    dev16 = sp.DeviceAlignment.from_table(O.unpack_table(g16["keys"], g16["probs"], 16), taxa=taxa_names(16))
    for al, code in ((dev, _lib.SP_METHOD_FLATTENING), (dev16, _lib.SP_METHOD_SUBFLATTENING), (dev, _lib.SP_METHOD_SUBFLATTENING)):
        full, st_full = batch.score_all_splits_shard(al, code, False, None, 0, 1)
        for world in (2, 3, 8):
            shards, total = batch.shard_layout(al.n_taxa, world)
            assert total == len(full)
            out = np.full(total, np.nan)
            for r in range(world):
                per = len(shards[r])
                buf = torch.zeros(batch.packed_width(per), dtype=torch.float64, device="cuda")
                n_loc = batch.score_all_splits_shard(al, code, False, None, r, world, scores_dev_ptr=buf.data_ptr(),
                                                     status_dev_ptr=buf.data_ptr() + per * 8)
                torch.cuda.synchronize()
                host = buf.cpu().numpy()
                assert n_loc == per and not np.any(host[per:].view(np.int32)[:per] & 3)
                out[shards[r]] = host[:per]
            assert np.array_equal(out, full), (code, world)
    assert np.abs(batch.score_all_splits_shard(dev, _lib.SP_METHOD_FLATTENING, False, None, 0, 1)[0] - g["scores"]).max() <= SCORE_TOL
    one_class, _ = batch.score_all_splits_shard(dev, _lib.SP_METHOD_FLATTENING, False, 4, 1, 3)
    assert np.array_equal(one_class, sp.score_all_splits(dev, size=4)[1::3])


def test_regress_stalled_block_is_not_accepted(sp):
    """Randomised sweep, seed 9100 (round 2): 8 taxa, 8 patterns, the 4|4 split below flattens to a 5 x 7 matrix with
    squared singular values 585.0, 433, 2, 1, 0.985.  The 4-wide block stalls on {1, 2, 3, 5} (the 4th direction is an
    isolated cell the start block barely sees, and 0.985 / 1 per half product never catches up); the stop rule's estimate of
    the block's smallest Ritz value came from inverse iteration started at (1, 1, 1, 1), which the smallest eigenvector had
    turned orthogonal to - the next eigenvalue came back, the gap guard passed and a score 2.4e-4 off was accepted as
    converged.  The estimate is certified now (S - 0.9 mu I positive definite): the split goes down the chain instead."""
    n = 8
    names = taxa_names(n)
    pats = ['0000000000000000', '0000000001000000', '0000000001000001', '0000010100000100', '0000010101000100',
            '0101000001010001', '0101010100010101', '0101010101010101']
    keys = np.array([int(p, 2) for p in pats], dtype=np.uint64)
    counts = np.array([24, 3, 1, 1, 1, 1, 12, 17], dtype=np.int64)
    a, b = [2, 5, 6, 7], [0, 1, 3, 4]
    split = (tuple(names[t] for t in a), tuple(names[t] for t in b))
    M = O.reduced_flattening_packed(keys, counts.astype(np.float64), n, a, b)[0]
    assert M.shape == (5, 7)
    want = O.dense_split_score(M)
    assert abs(want - 0.031038601176007503) < 1e-15
    dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=int(counts.sum()), taxa=names)
    for route in ("auto", "dense"):
        got, st = sp.score_splits(dev, [split], route=route, return_status=True)
        assert abs(got[0] - want) <= SCORE_TOL and (st[0] & 3) == 0, (route, got[0], want, hex(int(st[0])))
    # every split of the table, and the same table with counts beyond 16 bits (pieces of one cell in the lists)
    splits = list(sp.all_splits(names))
    for scale in (1, 70_000):
        d2 = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts * scale, n_sites=int(counts.sum()) * scale, taxa=names)
        got = sp.score_splits(d2, splits)
        for i, (x, y) in enumerate(splits):
            Mi = O.reduced_flattening_packed(keys, (counts * scale).astype(np.float64), n, [names.index(t) for t in x],
                                             [names.index(t) for t in y])[0]
            w = 0.0 if min(Mi.shape) <= 4 else O.dense_split_score(Mi)
            assert abs(got[i] - w) <= SCORE_TOL or abs(got[i] ** 2 - w ** 2) <= 4e-15, (scale, i, got[i], w)


def test_split_counts_are_reproducible(sp):
    """Counts >= 2^16 enter the sparse kernel's table as several rows of one cell.  Until late in round 2 every such piece
    stored its own value into the start block (the last writer won): 47 of 300 random tables of this kind gave scores
    that differed in their last bits between two runs in one process.  The cells are cleared and summed by exact fp64
    atomics now: three fresh alignments of every table must agree bit for bit."""
    rng = np.random.default_rng(77)
    for trial in range(60):
        n = int(rng.integers(4, 10))
        length = int(rng.choice([400, 2500, 20000]))
        keys, counts = _copy_mutate_table(rng, n, length, int(rng.choice([2, 3, 4])))
        counts = counts * int(rng.choice([300, 70_000]))
        names = taxa_names(n)
        splits = list(sp.all_splits(names))[:120]
        runs = []
        for _ in range(3):
            dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=int(counts.sum()), taxa=names)
            runs.append(sp.score_splits(dev, splits))
        assert np.array_equal(runs[0], runs[1], equal_nan=True) and np.array_equal(runs[0], runs[2], equal_nan=True), (trial, n, len(keys))


def test_regress_wide_block_close_pair(sp):
    """Two 7-taxon tables from the randomised sweeps of round 2 (seeds 37000 / 51000) whose flattenings have
    lambda_4 / lambda_5 = 1.0001 ... 1.0003 behind three dominant values.  Forced onto the big-table form, its 8-wide
    fallback block accepted sums whose 4th value sat on lambda_5 (the direction of lambda_4 still creeping up in 5th
    place) or that missed one direction of the pair altogether (amplitude 1e-8 in the guard columns: nothing moves for
    dozens of half products, every test passes, error bound included): scores 3e-6 off, status "converged".  The 5th
    value's reach is now the geometric tail of its own steps and the verdict has to last as long as such a direction needs
    to grow into view."""
    import os

    from tests.conftest import GOLDEN

    d = np.load(os.path.join(GOLDEN, "regress_wide_close_pair.npz"))
    for tag in ("a", "b"):
        n = int(d[tag + "_n"])
        names = taxa_names(n)
        keys, counts = d[tag + "_keys"], d[tag + "_counts"]
        splits = list(sp.all_splits(names))
        pick = [splits[int(i)] for i in d[tag + "_splits"]]
        dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=int(counts.sum()), taxa=names)
        got0 = sp.score_splits(dev, pick)                       # default chain
        dev.ctx.set_option("force_big", 1)
        try:
            got1, st1 = sp.score_splits(dev, pick, return_status=True)
        finally:
            dev.ctx.set_option("force_big", 0)
        assert np.abs(got0 - d[tag + "_want"]).max() <= SCORE_TOL, (tag, got0, d[tag + "_want"])
        assert np.abs(got1 - d[tag + "_want"]).max() <= SCORE_TOL and not np.any(st1 & 3), (tag, got1, d[tag + "_want"], st1)


def test_plan_steps_entry(sp, golden):
    """sp_score_plan_steps (ABI 3): n_steps complete passes from one host call, each into its own output slot - every slot
    bit-identical to a single sp_score_plan_async pass (scores AND status words), strides checked."""
    import ctypes as C
    import torch
    from splitp_amd import batch

    g = golden("n10_L100k")
    names = taxa_names(10)
    dev = sp.DeviceAlignment.from_table(O.unpack_table(g["keys"], g["probs"], 10), taxa=names)
    taxa_arr, a_arr = sp.encode_all_splits(10)
    plan = batch.SplitPlan(dev.ctx, taxa_arr, a_arr, 10)
    s_count, steps = len(a_arr), 3
    width = batch.packed_width(s_count)
    one = torch.zeros(width, dtype=torch.float64, device="cuda")
    many = torch.zeros(steps * width, dtype=torch.float64, device="cuda")
    dev.ctx.sync_stream_with_torch()
    batch.score_plan_async(dev.ctx, [dev], plan, one.data_ptr(), one.data_ptr() + s_count * 8)
    batch.score_plan_steps(dev.ctx, [dev], plan, steps, many.data_ptr(), width * 8, many.data_ptr() + s_count * 8, width * 8)
    dev.ctx.synchronize()
    torch.cuda.synchronize()
    ref = one.cpu().numpy()
    got = many.cpu().numpy().reshape(steps, width)
    assert np.abs(ref[:s_count] - g["scores"]).max() <= SCORE_TOL
    for s_ in range(steps):
        assert np.array_equal(got[s_], ref)                                  # scores and packed status words, bit for bit
    st = ref[s_count:].view(np.int32)[:s_count]
    assert not np.any(st & 3) and np.all((st >> 8) >= 3)
    lib = dev.ctx._lib
    h = (C.c_void_p * 1)(dev.handle.value)
    bad = lib.sp_score_plan_steps(dev.ctx.handle, h, 1, plan.handle, 2, C.c_void_p(many.data_ptr()), 8, C.c_void_p(many.data_ptr()), 8)
    assert bad == 1 and b"step strides" in lib.sp_last_error()                # SP_EINVAL: slots would overlap
    assert lib.sp_score_plan_steps(dev.ctx.handle, h, 1, plan.handle, 0, C.c_void_p(many.data_ptr()), 0, C.c_void_p(many.data_ptr()), 0) == 0


def test_distributed_nccl_world_size_one(sp, golden):
    """The RCCL code path with the real kernels: `nccl` initialised in-process at world size 1 on this
    box's one GPU; score_splits(distributed=True) and score_all_splits(distributed=True) - kernels writing scores and
    status straight into the all-gather's send buffer (batch.py nccl branches) - equal the undistributed calls bit for
    bit, status words included.  Ordering contract: reference splits.py:39-59."""
    import os
    import socket
    import torch
    import torch.distributed as dist

    if dist.is_initialized():
        pytest.skip("a process group is already initialised in this process")
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        assert dist.get_backend() == "nccl"
        g = golden("n10_L100k")
        names = taxa_names(10)
        dev = sp.DeviceAlignment.from_table(O.unpack_table(g["keys"], g["probs"], 10), taxa=names)
        splits = list(sp.all_splits(names))
        plain, plain_st = sp.score_splits(dev, splits, return_status=True)
        dsc, dst = sp.score_splits(dev, splits, distributed=True, return_status=True)
        assert np.array_equal(dsc, plain) and np.array_equal(dst & 3, plain_st & 3) and np.array_equal(dst >> 8, plain_st >> 8)
        assert np.abs(dsc - g["scores"]).max() <= SCORE_TOL
        asc, ast = sp.score_all_splits(dev, distributed=True, return_status=True)
        assert np.array_equal(asc, plain) and np.array_equal(ast, plain_st)
        # subflattening route, 16 taxa (the partition of BASELINE configs 3 / 4: shards enumerated on the device)
        g16 = golden("n16_L4k")
        names16 = taxa_names(16)
        dev16 = sp.DeviceAlignment.from_table(O.unpack_table(g16["keys"], g16["probs"], 16), taxa=names16)
        want, want_st = sp.score_all_splits(dev16, method=sp.Method.subflattening, return_status=True)
        got, got_st = sp.score_all_splits(dev16, method=sp.Method.subflattening, distributed=True, return_status=True)
        assert len(got) == 32751 and np.array_equal(got, want) and np.array_equal(got_st, want_st)
        some = splits[::7]
        a = sp.score_splits(dev, some, method=sp.Method.subflattening, distributed=True)
        assert np.array_equal(a, sp.score_splits(dev, some, method=sp.Method.subflattening))
    finally:
        dist.destroy_process_group()


def test_randomised_sweep_slice(sp):
    """A bounded slice of the randomised sweeps of tools/gpu_fuzz_long.py inside the suite (the six defects of
    round 2 were all found by those tools, none by pytest): fixed seeds - among them the families that exposed the false
    acceptance (9100), the wide-block verdicts (37000, 51000) -, tables unlike anything a tree produces (4 - 12 taxa,
    2 - 4 letters, counts up to 70 000 x), every split against the oracle on the default chain, on the hand-back chain
    with the LDS capped at 30 KB and 12 KB (lists-in-global / all-global forms) and on the forced big-table form.
    Gate as in the tools: score within 1e-10, or 1 - top4/trace within 8e-15 (the fp64 floor, decides scores < 2e-5)."""
    import time
    from splitp_amd.device import Context, current_device

    def close(a, b):
        return abs(a - b) <= 1e-10 or abs(a * a - b * b) <= 8e-15

    t0 = time.time()
    checked = 0
    variants = []
    for vname, opts in (("default", {}), ("lds_cap 30 KB", {"lds_cap": 30000}), ("lds_cap 12 KB", {"lds_cap": 12000}),
                        ("big-table form", {"force_big": 1})):
        ctx = Context(current_device())          # one context per variant: options are per context
        for key, val in opts.items():
            ctx.set_option(key, val)
        variants.append((vname, ctx))
    for seed0, ntr in ((9100, 16), (37000, 12), (51000, 12), (2026, 16)):
        rng = np.random.default_rng(seed0)
        for trial in range(ntr):
            n = int(rng.integers(4, 13))
            length = int(rng.choice([10, 60, 400, 2500, 20000]))
            letters = int(rng.choice([2, 3, 4, 4]))
            keys, counts = _copy_mutate_table(rng, n, length, letters)
            if trial % 7 == 0:
                counts = counts * int(rng.choice([300, 70_000]))
            names = taxa_names(n)
            if n <= 6:
                splits = list(sp.all_splits(names))
            else:
                splits = []
                for _ in range(10):
                    k = int(rng.integers(2, n - 1))
                    left = sorted(rng.choice(n, size=k, replace=False).tolist())
                    splits.append((tuple(names[t] for t in left), tuple(names[t] for t in range(n) if t not in left)))
            want = []
            for spl in splits:
                M = O.reduced_flattening_packed(keys, counts.astype(np.float64), n, [names.index(t) for t in spl[0]],
                                                [names.index(t) for t in spl[1]])[0]
                w = None if min(M.shape) > 400 else (0.0 if min(M.shape) <= 4 else O.dense_split_score(M))
                want.append(0.0 if (w is not None and np.isnan(w)) else w)
            which = variants if trial % 2 == 0 else variants[:1]
            for vname, ctx in which:
                dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=int(counts.sum()), taxa=names, ctx=ctx)
                got, st = sp.score_splits(dev, splits, return_status=True)
                for i, w in enumerate(want):
                    if w is None:
                        continue
                    checked += 1
                    assert close(w, got[i]) or (st[i] & 1), (vname, seed0, trial, n, length, letters, i, w, got[i], hex(int(st[i])))
    assert checked >= 1000, checked
    assert time.time() - t0 < 120, "the sweep slice is meant to stay within a minute or two"
