"""CPU: the C-ABI library loads and exports every symbol include/splitp_hip.h declares (no compute
calls without a GPU), and the host-side logic (packing, count inference, split encoding / ordering,
sharding plan, error behaviour without a GPU)."""
import os
import re

import numpy as np
import pytest

import splitp_amd as sp
from oracle import splitp_oracle as O
from splitp_amd import _lib, batch, device
from splitp_amd import synthetic as syn
from tests.conftest import ROOT, taxa_names


def _built():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()


def test_library_exports_every_declared_symbol():
    _built()
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "splitp_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(sp_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert sorted(_lib.SYMBOLS) == declared
    assert lib.sp_abi_version() == _lib.SP_ABI_VERSION == 4
    assert isinstance(lib.sp_last_error(), bytes)


def test_no_gpu_means_loud_failure():
    _built()
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(sp.SplitPDeviceError):
        sp.flattening("01|23", {"ATCG": 0.4, "GATC": 0.2, "CGAT": 0.2, "TCGA": 0.2})
    with pytest.raises(sp.SplitPDeviceError):
        sp.split_score(np.eye(6))
    with pytest.raises(sp.SplitPDeviceError):
        sp.subflattening("01|23", {"ATCG": 1.0})


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "splitp_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src and "oracle." not in src, fn


def test_api_surface_matches_reference_signatures():
    import inspect

    assert str(inspect.signature(sp.flattening)) == "(split, pattern_probabilities, flattening_format=<FlatFormat.sparse: 'sparse'>)"
    assert str(inspect.signature(sp.subflattening)) == "(split, pattern_probabilities, data=None)"
    assert str(inspect.signature(sp.split_score)) == ("(matrix, return_singular_values=False, "
                                                       "force_frob_norm_on_dense=False, data_table_for_frob_norm=None)")
    assert [m.name for m in sp.FlatFormat] == ["sparse", "reduced"] and sp.FlatFormat.sparse.value == "sparse"
    assert sp.constants.DNA_state_space == ("A", "C", "G", "T")
    assert sp.constants.DNA_state_space_dict == {"A": 0, "C": 1, "G": 2, "T": 3}
    assert sp.constructions.flattening is sp.flattening and sp.phylogenetics.split_score is sp.split_score


def test_pack_patterns_and_keys():
    pats = ["ACGT", "TTTT", "AAAA", "GATC"]
    keys, n = device.pack_patterns(pats)
    assert n == 4 and keys.dtype == np.uint64
    assert keys.tolist() == [O.index_of(p) for p in pats] == [27, 255, 0, 141]
    with pytest.raises(KeyError):
        device.pack_patterns(["ACGN"])
    with pytest.raises(ValueError):
        device.pack_patterns(["ACG", "ACGT"])
    k0, n0 = device.pack_patterns([])
    assert len(k0) == 0
    # round trip through the oracle's packer on a simulated table
    sites = syn.simulate_sites(10, 3000, 0.05, seed=4)
    keys, counts = syn.pattern_table(sites)
    table = syn.table_as_dict(keys, counts, 10)
    k2, _ = device.pack_patterns(list(table.keys()))
    assert np.array_equal(k2, keys) and np.array_equal(O.pack_table(table)[0], keys)


def test_infer_counts():
    sites = syn.simulate_sites(8, 5000, 0.05, seed=9)
    keys, counts = syn.pattern_table(sites)
    w = counts / 5000.0
    got = device.infer_counts(w)
    assert got is not None
    cnt, n = got
    assert np.array_equal(cnt / float(n), w)          # equivalent (counts, N): same values bit for bit
    assert device.infer_counts(np.array([0.4, 0.2, 0.2, 0.2]))[1] in (5, 10)
    rng = np.random.default_rng(0)
    p = rng.random(50); p /= p.sum()
    assert device.infer_counts(p) is None             # exact probabilities are not count-derived
    assert device.infer_counts(np.array([])) is None
    assert device.infer_counts(np.array([0.0, 0.0])) is None


def test_all_splits_order_and_counts():
    for n in (4, 6, 10):
        names = taxa_names(n)
        mine = list(sp.all_splits(names))
        assert mine == list(O.all_splits(names))
        assert len(mine) == 2 ** (n - 1) - n - 1
        assert all(s[0][0] == names[0] for s in mine)     # taxa[0] always on the left (splits.py:55-56)
    assert len(list(sp.all_splits(taxa_names(10), size=5))) == 126
    assert next(sp.all_splits(taxa_names(6), string_format=True)) == "01|2345"
    assert len(list(sp.all_splits(taxa_names(4), trivial=True))) == 7


def test_balanced_tree_matches_reference_topology():
    assert syn.balanced_tree(10) == ((((0, 1), 2), (3, 4)), ((5, 6), (7, (8, 9))))
    assert syn.balanced_tree(4) == ((0, 1), (2, 3))
    sets = syn.tree_splits(syn.balanced_tree(10), 10)   # both sides of the root edge appear: 8 sets, 7 splits
    assert len({min(s, frozenset(range(10)) - s, key=sorted) for s in sets}) == 7
    assert syn.taxa_names(12)[:12] == ["0", "1", "2", "3", "4", "5", "6", "7", "8", "9", "A", "B"]


def test_encode_splits_and_resolve():
    names = taxa_names(6)

    class T(dict):
        pass

    t = T()
    t.taxa = tuple(names)
    taxa_arr, a_arr = batch.encode_splits(["01|2345", (("5", "0"), ("1", "2", "3", "4"))], t, 6)
    assert a_arr.tolist() == [2, 2]
    assert taxa_arr.tolist() == [[0, 1, 2, 3, 4, 5], [5, 0, 1, 2, 3, 4]]
    with pytest.raises(ValueError):
        batch.encode_splits(["01|234"], t, 6)
    oa, ob = device.resolve_split(({0, 2}, {1, 3}), {}, 4)       # int taxa, sets (reference tests' form)
    assert sorted(oa.tolist()) == [0, 2] and sorted(ob.tolist()) == [1, 3]


def test_encode_all_splits_matches_all_splits():
    """The NumPy enumeration of the candidate splits = encode_splits(all_splits(taxa)) (reference splits.py:39-59 order),
    for every taxon count and option."""
    class T:
        pass

    for n in range(2, 12):
        names = taxa_names(n)
        t = T()
        t.taxa = tuple(names)
        for kw in ({}, {"trivial": True}, {"size": max(1, n // 2)}, {"size": 1}):
            lst = list(sp.all_splits(names, **kw))
            ta, aa = batch.encode_all_splits(n, **kw)
            assert len(aa) == len(lst), (n, kw)
            if lst:
                tb, ab = batch.encode_splits(lst, t, n)
                assert np.array_equal(ta, tb) and np.array_equal(aa, ab), (n, kw)


def test_shard_plan_is_balanced_and_complete():
    a = np.array([len(s[0]) for s in sp.all_splits(taxa_names(10))], dtype=np.int32)
    costs = batch.split_costs(a, 10, _lib.SP_METHOD_FLATTENING)
    for world in (1, 2, 4, 8):
        shards = batch.shard_indices(costs, world)
        allidx = np.sort(np.concatenate(shards))
        assert np.array_equal(allidx, np.arange(501))
        loads = [costs[s].sum() for s in shards]
        assert max(loads) / min(loads) < 1.05          # every rank gets an equal share of every class
        assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1


def test_simulation_host_logic():
    """Host side of the device simulator (no GPU): tree -> parents-first arrays for both tree forms, and the closed-form
    Jukes-Cantor matrix against expm of the reference's normalised rate matrix (model.py:14-16, :49-58, :66-75)."""
    import numpy as np
    import scipy.linalg
    from splitp_amd import simulation as sim
    from splitp_amd import synthetic as syn

    jc = sim.JukesCantor()
    q = np.full((4, 4), 1.0 / 3.0)
    np.fill_diagonal(q, -1.0)                      # JC rate matrix scaled to one expected substitution per unit time
    for t in (0.0, 0.05, 0.7, 3.0):
        np.testing.assert_allclose(jc.transition_matrix(t), scipy.linalg.expm(t * q), rtol=0, atol=1e-15)
    parent, leaf, trans, taxa = sim.tree_arrays(syn.balanced_tree(10), jc, 0.05)
    assert len(parent) == 19 and parent[0] == -1 and all(parent[i] < i for i in range(1, 19))
    assert sorted(int(x) for x in leaf if x >= 0) == list(range(10)) and trans.shape == (19, 4, 4)
    assert taxa == syn.taxa_names(10)
    np.testing.assert_allclose(trans[1:].sum(axis=1), 1.0, atol=1e-15)      # columns are distributions
    # the reference's Phylogeny, as far as the simulator looks at it
    import networkx as nx
    g = nx.DiGraph()
    g.add_edges_from([("r", "x"), ("r", "C"), ("x", "A"), ("x", "B")])
    for node in g.nodes:
        g.nodes[node]["branch_length"] = 0.2

    class Phylo:
        networkx_graph = g
        taxa = ["A", "B", "C"]

    parent, leaf, trans, taxa = sim.tree_arrays(Phylo(), jc)
    assert parent.tolist()[0] == -1 and sorted(x for x in leaf.tolist() if x >= 0) == [0, 1, 2] and taxa == ["A", "B", "C"]
    np.testing.assert_allclose(trans[1], jc.transition_matrix(0.2))
    with pytest.raises(ValueError):
        sim.tree_arrays(syn.balanced_tree(4), None, 0.1)          # nested tuples carry no matrices
    with pytest.raises(ValueError):
        sim.tree_arrays(syn.balanced_tree(4), jc)                 # ... and no branch lengths


def test_table_fingerprint_sees_every_value():
    """the drop-in path's table cache must miss on ANY in-place edit, also one that keeps the keys
    and the total mass (values swapped, counts moved between patterns of a fixed-length dict)."""
    from splitp_amd import device

    table = {"ACGT": 0.25, "AAAA": 0.30, "CCCC": 0.20, "GGTT": 0.25}
    fp = device._fingerprint(table)
    assert device._fingerprint(table) == fp
    table["ACGT"], table["AAAA"] = table["AAAA"], table["ACGT"]            # same keys, same sum
    assert device._fingerprint(table) != fp
    counts = {"ACGT": 5 / 20, "AAAA": 10 / 20, "CCCC": 5 / 20}
    fp = device._fingerprint(counts)
    counts["ACGT"], counts["CCCC"] = 3 / 20, 7 / 20                        # a bootstrap refill: 2 counts moved
    assert device._fingerprint(counts) != fp


def test_flattening_content_check_detects_edits():
    """flattening(..., reduced) hands out a writable array (reference constructions.py:51); the remembered origin is
    dropped as soon as the contents differ from what the device produced."""
    from splitp_amd.constructions import Flattening, _content_check, flattening_origin

    base = np.arange(12, dtype=np.float64).reshape(3, 4) / 7.0
    F = base.copy().view(Flattening)
    F._sp_origin = ("al", np.array([0, 1]), np.array([2, 3]))
    F._sp_check = _content_check(F)
    assert F.flags.writeable and flattening_origin(F) is F._sp_origin
    assert flattening_origin(F.copy()) is None and flattening_origin(F * 1.0) is None and flattening_origin(F[:, 1:]) is None
    F[1, 2], F[2, 1] = F[2, 1], F[1, 2]                                     # a swap keeps the plain sum
    assert flattening_origin(F) is None and F._sp_origin is None
    G = base.copy().view(Flattening)
    G._sp_origin = ("al", None, None)
    G._sp_check = _content_check(G)
    G /= G.sum()
    assert flattening_origin(G) is None


def test_score_prefetch_credit_logic(monkeypatch):
    """Host logic of the drop-in loop's score prefetch (constructions._prefetch_score / take_prefetched_score), no GPU: a
    table's flattenings are prefetched only after split_score has consumed one, a caller who stops scoring costs two
    launches, a prefetch that cannot be enqueued is not an error, an overtaken slot sends the caller to the synchronous call."""
    from types import SimpleNamespace

    from splitp_amd import constructions as K
    from splitp_amd.constructions import Flattening

    issued = []

    class FakePrefetch:
        def __init__(self, device):
            self.device = device
            self.fail = False

        def issue(self, al, oa, ob):
            if self.fail:
                raise RuntimeError("cannot enqueue")
            issued.append((tuple(oa), tuple(ob)))
            return (self, len(issued) - 1, None, None)

        def collect(self, al, pending):
            return None if pending[1] == 0 else np.float64(pending[1])

    monkeypatch.setattr(K, "_ScorePrefetch", FakePrefetch)
    monkeypatch.setattr(K, "_prefetchers", {})
    al = SimpleNamespace(ctx=SimpleNamespace(device=3))
    oa, ob = np.array([0, 1], dtype=np.int32), np.array([2, 3], dtype=np.int32)
    assert K._prefetch_score(al, oa, ob) is None and not issued            # nobody has scored a flattening of this table yet
    F = np.zeros((2, 2)).view(Flattening)
    assert K.take_prefetched_score(F, al) is None and al._sp_prefetch_credit == 2
    p0 = K._prefetch_score(al, oa, ob)
    p1 = K._prefetch_score(al, oa, ob)
    assert p0 is not None and p1 is not None and len(issued) == 2
    assert K._prefetch_score(al, oa, ob) is None and len(issued) == 2       # credit used up: two unused launches, no more
    F._sp_pending = p1
    assert K.take_prefetched_score(F, al) == 1.0 and F._sp_pending is None and al._sp_prefetch_credit == 2
    F._sp_pending = p0                                                      # (the fake's slot 0 counts as overtaken)
    assert K.take_prefetched_score(F, al) is None and F._sp_pending is None
    K._prefetchers[3].fail = True
    assert K._prefetch_score(al, oa, ob) is None and al._sp_prefetch_credit == 0
    monkeypatch.setattr(K, "PREFETCH_SCORES", False)
    al._sp_prefetch_credit = 2
    assert K._prefetch_score(al, oa, ob) is None and al._sp_prefetch_credit == 2


def test_new_kernels_keep_nothing_in_scratch():
    """The compiler's own resource remarks (tools/kernel_resources.py, no GPU): the round-4 kernels - the certified 4-wide eigen
    kernel of the dense route and the direct solver - spill no vector register and use no scratch memory, and the committed
    table of all kernels (profiles/r04_kernel_resources.json) says the same of the headline kernel."""
    import json
    import shutil
    import sys

    if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources as kr

    for fn in ("eig4.hip", "finish.hip"):
        table = kr.resources(os.path.join(ROOT, "splitp_amd", "csrc", fn))
        assert table, fn
        for name, res in table.items():
            assert res.get("scratch_bytes_per_lane") == 0 and res.get("vgpr_spills") == 0, (fn, name, res)
    committed = json.load(open(os.path.join(ROOT, "profiles", "r04_kernel_resources.json")))
    fast = committed["sparse.hip"]["k_sparse_score"]
    assert fast["scratch_bytes_per_lane"] == 0 and fast["vgpr_spills"] == 0
    big = [v for k, v in committed["sparse_big.hip"].items() if "k_sparse_big" in k and "false" in k]
    assert big and all(v["vgpr_spills"] <= 32 for v in big)     # (round 3: 585; the wide fallback is a kernel of its own now)


def test_hand_issued_scalar_loads_have_no_hazard():
    """subflat.hip fetches its Sturm table with inline-asm s_load_dwordx8 into registers the hardware writes asynchronously
    (ADVICE r3): on every control-flow path from such a load to its s_waitcnt lgkmcnt(0) no instruction may touch the
    destination registers - checked on the device assembly of the current compiler (tools/check_sload_hazard.py)."""
    import shutil
    import sys

    if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_sload_hazard as chk

    n, bad = chk.check(os.path.join(ROOT, "splitp_amd", "csrc", "subflat.hip"))
    assert n >= 16 and not bad, bad[:5]


def test_install_as_splitp_alias():
    """`import splitp` after splitp_amd.install_as_splitp(): the reference's import lines work unchanged."""
    import importlib
    import sys

    saved = {k: v for k, v in sys.modules.items() if k == "splitp" or k.startswith("splitp.")}
    try:
        for k in saved:
            del sys.modules[k]
        sp.install_as_splitp()
        import splitp                                    # noqa: F401
        from splitp.constructions import flattening      # noqa: F401
        from splitp.phylogenetics import split_score     # noqa: F401
        from splitp import FlatFormat, all_splits        # noqa: F401
        assert splitp is sp and flattening is sp.flattening and split_score is sp.split_score
        assert importlib.import_module("splitp.enums").FlatFormat is sp.FlatFormat
    finally:
        for k in [k for k in sys.modules if k == "splitp" or k.startswith("splitp.")]:
            del sys.modules[k]
        sys.modules.update(saved)
