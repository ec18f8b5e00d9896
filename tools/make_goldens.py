#!/usr/bin/env python3
"""Generate golden input/output vectors from the real reference (js51/SplitP).

Runs ONLY in the build container, where the reference is mounted read-only at
/root/reference.  It imports the reference's `splitp` package (pure Python) and
writes *data only* (inputs + expected outputs) as small .npz fixtures under
tests/golden/.  Nothing of the reference's source text is stored.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_goldens.py [--only NAME]

The fixtures pin the oracle (oracle/splitp_oracle.py) and, through it, the HIP
path.  The GPU box never runs this script (no /root/reference there).
"""
import argparse
import os
import random
import sys
import time

import numpy as np

REF = os.environ.get("SPLITP_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

STATE = {"A": 0, "C": 1, "G": 2, "T": 3}


def pack(table):
    """dict pattern->prob  ->  (keys uint64 [taxon 0 most significant], probs f64), dict order kept."""
    keys = np.empty(len(table), dtype=np.uint64)
    probs = np.empty(len(table), dtype=np.float64)
    for i, (p, v) in enumerate(table.items()):
        k = 0
        for ch in p:
            k = k * 4 + STATE[ch]
        keys[i] = k
        probs[i] = v
    return keys, probs


def split_to_orders(split, taxa):
    idx = {t: i for i, t in enumerate(taxa)}
    return (np.array([idx[t] for t in split[0]], dtype=np.int32),
            np.array([idx[t] for t in split[1]], dtype=np.int32))


def left_mask(split, taxa):
    idx = {t: i for i, t in enumerate(taxa)}
    m = 0
    for t in split[0]:
        m |= 1 << idx[t]
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    if not os.path.isdir(REF):
        print("reference not present; nothing to do")
        return 0
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import splitp  # the REAL reference
    from splitp.constructions import flattening, subflattening
    from splitp.phylogenetics import split_score
    from splitp.enums import FlatFormat
    import scipy, networkx
    os.makedirs(OUT, exist_ok=True)
    versions = f"numpy {np.__version__} scipy {scipy.__version__} networkx {networkx.__version__} splitp {open(os.path.join(REF, 'VERSION')).read().strip()}"

    def want(name):
        return args.only is None or args.only == name

    # ------------------------------------------------------------------ 4 taxa
    if want("ref4"):
        table = {"ATCG": 2 / 5, "GATC": 1 / 5, "CGAT": 1 / 5, "TCGA": 1 / 5}
        splits = [({0, 1}, {2, 3}), ({0, 2}, {1, 3}), ({0, 3}, {1, 2})]
        keys, probs = pack(table)
        out = dict(keys=keys, probs=probs, n_taxa=4, versions=versions)
        for i, s in enumerate(splits):
            out[f"sparse_{i}"] = np.asarray(flattening(s, table).todense())
            out[f"reduced_{i}"] = flattening(s, table, FlatFormat.reduced)
            out[f"subflat_{i}"] = subflattening(s, table)
            out[f"orderA_{i}"] = np.array(list(s[0]), dtype=np.int32)   # iteration order of the set
            out[f"orderB_{i}"] = np.array(list(s[1]), dtype=np.int32)
            out[f"score_sub_{i}"] = split_score(out[f"subflat_{i}"])
            out[f"score_red_{i}"] = split_score(out[f"reduced_{i}"])   # 4x4, rank<=4 -> 0 or nan
        # string-form split and a non-sorted taxon order inside a half
        out["sparse_str"] = np.asarray(flattening("10|32", table).todense())
        out["reduced_str"] = flattening("10|32", table, FlatFormat.reduced)
        # subflattening("10|32", plain dict) raises KeyError in the reference (the '|' is counted as a
        # taxon, constructions.py:114-117); it works when the table carries .taxa:
        from splitp.alignment import Alignment
        out["subflat_str"] = subflattening("10|32", Alignment(dict(table), taxa=("0", "1", "2", "3")))
        np.savez_compressed(os.path.join(OUT, "ref4.npz"), **out)
        print("ref4 done")

    # ------------------------------------------------------- 10 taxa alignments
    def do_n10(name, L, seed, n_full, n_sparse, n_sub):
        t0 = time.time()
        random.seed(seed)
        np.random.seed(seed)
        tree = splitp.trees.balanced_newick_tree(10, 0.05)
        model = splitp.model.GTR.JukesCantor(1 / 2)
        table = splitp.generate_alignment(tree, model, L)
        t_gen = time.time() - t0
        taxa = tree.taxa
        splits = list(splitp.all_splits(tree))
        true_splits = set()
        for s in tree.splits():
            true_splits.add(frozenset(s[0])); true_splits.add(frozenset(s[1]))
        keys, probs = pack(table)
        masks = np.array([left_mask(s, taxa) for s in splits], dtype=np.uint32)
        is_true = np.array([frozenset(s[0]) in true_splits for s in splits], dtype=bool)
        scores = np.empty(len(splits)); shapes = np.empty((len(splits), 2), dtype=np.int32)
        t_flat = t_svd = 0.0
        out = dict(keys=keys, probs=probs, n_taxa=10, L=L, seed=seed, masks=masks, is_true=is_true,
                   taxa=np.array(list(taxa)), versions=versions)
        full_ids = []
        for k in (2, 3, 4, 5):
            ids = [i for i, s in enumerate(splits) if len(s[0]) == k or len(s[1]) == k and len(s[0]) > len(s[1])]
            ids = [i for i, s in enumerate(splits) if min(len(s[0]), len(s[1])) == k]
            full_ids += ids[:: max(1, len(ids) // n_full)][:n_full]
        for i, s in enumerate(splits):
            t1 = time.time()
            F = flattening(s, table, FlatFormat.reduced)
            t2 = time.time()
            scores[i] = split_score(F)
            t3 = time.time()
            t_flat += t2 - t1; t_svd += t3 - t2
            shapes[i] = F.shape
            if i in full_ids:
                out[f"reduced_{i}"] = F
        out["scores"] = scores; out["shapes"] = shapes; out["full_ids"] = np.array(full_ids)
        # sparse (default) path on a sample
        sp_ids = list(range(0, len(splits), max(1, len(splits) // n_sparse)))[:n_sparse]
        sp_scores = []
        for i in sp_ids:
            Fs = flattening(splits[i], table)
            sp_scores.append(split_score(Fs))
            if i == sp_ids[0]:
                coo = Fs.tocoo()
                out["sparse0_rows"] = coo.row; out["sparse0_cols"] = coo.col; out["sparse0_vals"] = coo.data
                out["sparse0_shape"] = np.array(Fs.shape)
        out["sparse_ids"] = np.array(sp_ids); out["sparse_scores"] = np.array(sp_scores)
        # subflattenings on a sample
        sub_ids = []
        for k in (2, 3, 4, 5):
            ids = [i for i, s in enumerate(splits) if min(len(s[0]), len(s[1])) == k]
            sub_ids += ids[:: max(1, len(ids) // n_sub)][:n_sub]
        data = {}
        for i in sub_ids:
            S = subflattening(splits[i], table, data)
            out[f"subflat_{i}"] = S
            out[f"subscore_{i}"] = split_score(S)
        out["sub_ids"] = np.array(sub_ids)
        out["timing"] = np.array([t_gen, t_flat, t_svd])
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
        print(f"{name}: D={len(table)} gen {t_gen:.1f}s flat {t_flat:.1f}s svd {t_svd:.1f}s "
              f"-> {len(splits)/(t_flat+t_svd):.2f} splits/s")

    if want("n10_L10k"):
        do_n10("n10_L10k", 10_000, 0, n_full=1, n_sparse=8, n_sub=2)
    if want("n10_L100k"):
        do_n10("n10_L100k", 100_000, 12345, n_full=1, n_sparse=4, n_sub=1)

    # --------------------------------------------------- 16 taxa subflattening
    if want("n16_L4k"):
        random.seed(7)
        tree = splitp.trees.balanced_newick_tree(16, 0.05)
        model = splitp.model.GTR.JukesCantor(1 / 2)
        table = splitp.generate_alignment(tree, model, 4000)
        taxa = tree.taxa
        keys, probs = pack(table)
        out = dict(keys=keys, probs=probs, n_taxa=16, L=4000, versions=versions, taxa=np.array(list(taxa)))
        import itertools
        gen = splitp.all_splits(tree)
        splits = []
        for size in (2, 5, 8):
            g = splitp.all_splits(tree, size=size)
            ss = list(itertools.islice(g, 0, 4000, 1333))
            splits += ss
        # plus the true splits of the tree (non-trivial)
        for s in tree.splits():
            if min(len(s[0]), len(s[1])) >= 2:
                splits.append((tuple(sorted(s[0], key=taxa.index)), tuple(sorted(s[1], key=taxa.index))))
                if len(splits) >= 14:
                    break
        data = {}
        for i, s in enumerate(splits):
            oa, ob = split_to_orders(s, taxa)
            out[f"orderA_{i}"] = oa; out[f"orderB_{i}"] = ob
            S = subflattening(s, table, data)
            out[f"subflat_{i}"] = S
            out[f"subscore_{i}"] = split_score(S)
        out["n_splits"] = len(splits)
        np.savez_compressed(os.path.join(OUT, "n16_L4k.npz"), **out)
        print("n16_L4k done, D =", len(table), "splits", len(splits))

    # ------------------------------------------------- erickson_SVD (caller of the path)
    if want("erickson"):
        import json
        g = np.load(os.path.join(OUT, "n10_L10k.npz"))
        table = {}
        for k, v in zip(g["keys"].tolist(), g["probs"].tolist()):
            table["".join("ACGT"[(k >> (2 * (9 - t))) & 3] for t in range(10))] = v
        from splitp.phylogenetics import erickson_SVD
        import splitp as sp_ref
        res = {}
        t0 = time.time()
        res["flattening"] = erickson_SVD(table, method=sp_ref.Method.flattening)
        print("erickson flattening", time.time() - t0)
        t0 = time.time()
        res["subflattening"] = erickson_SVD(table, method=sp_ref.Method.subflattening)
        print("erickson subflattening", time.time() - t0)
        with open(os.path.join(OUT, "erickson_n10_L10k.json"), "w") as f:
            json.dump({k: [[list(map(str, side)) for side in s] for s in v] for k, v in res.items()}, f)
        print("erickson done", res["flattening"])

    # ------------------------------------- mutual-information score (phylogenetics.py:364-373) + its erickson_SVD
    if want("divergence"):
        import json
        from splitp.phylogenetics import erickson_SVD, flattening_rank_1_approximation_divergence
        import splitp as sp_ref
        g = np.load(os.path.join(OUT, "n10_L10k.npz"))
        names = [str(np.base_repr(i, base=max(i + 1, 2))) for i in range(10)]
        table = {}
        for k, v in zip(g["keys"].tolist(), g["probs"].tolist()):
            table["".join("ACGT"[(k >> (2 * (9 - t))) & 3] for t in range(10))] = v
        ids = list(range(0, 501, 12))
        vals = []
        t0 = time.time()
        for i in ids:
            m = int(g["masks"][i])
            split = (tuple(names[t] for t in range(10) if (m >> t) & 1), tuple(names[t] for t in range(10) if not (m >> t) & 1))
            flat = flattening(split, table, FlatFormat.reduced)
            vals.append(float(flattening_rank_1_approximation_divergence(flat)))
        print("divergence of", len(ids), "splits", time.time() - t0)
        # the 4-taxon reference table too (dense 4 x 4 reduced flattenings)
        ref4 = {"ATCG": 2 / 5, "GATC": 1 / 5, "CGAT": 1 / 5, "TCGA": 1 / 5}
        v4 = [float(flattening_rank_1_approximation_divergence(flattening(s, ref4, FlatFormat.reduced)))
              for s in ((("0", "1"), ("2", "3")), (("0", "2"), ("1", "3")), (("0", "3"), ("1", "2")))]
        t0 = time.time()
        tree = erickson_SVD(table, method=sp_ref.Method.mutual_information)
        print("erickson mutual_information", time.time() - t0)
        with open(os.path.join(OUT, "divergence_n10_L10k.json"), "w") as f:
            json.dump({"versions": versions, "split_ids": ids, "divergence": vals, "ref4": v4,
                       "erickson_mutual_information": [[list(map(str, side)) for side in s] for s in tree]}, f)

    # ------------------------------------------------- extras (round 2): simulator distribution, banned patterns,
    # rank-k approximation, frobenius_norm, restatement/reference wall-time ratio
    if want("extras"):
        import json
        repo = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        if repo not in sys.path:
            sys.path.append(repo)           # our own package / oracle: used to lay the reference's trees out as arrays
        from splitp import simulation as rsim
        from splitp.constructions import sparse_flattening_with_banned_patterns as sfb
        from splitp.phylogenetics import flattening_rank_k_approximation
        from splitp.matrix import frobenius_norm
        from splitp_amd import simulation as own_sim
        from oracle import splitp_oracle as O
        out = dict(versions=versions)
        model = splitp.model.GTR.JukesCantor(1 / 2)
        # (a) exact pattern probabilities (simulation.py:58-83) of a 4-taxon balanced tree and an unbalanced 5-taxon tree
        for tag, tree in (("sim4", splitp.trees.balanced_newick_tree(4, 0.1)),
                          ("sim5", splitp.Phylogeny("((A:0.1,B:0.2):0.05,(C:0.3,(D:0.1,E:0.15):0.07):0.02);"))):
            probs = rsim.get_pattern_probabilities(tree, model)
            parent, leaf, trans, taxa = own_sim.tree_arrays(tree, model)
            n = len(taxa)
            dense = np.zeros(4 ** n)
            for pat, v in probs.items():
                k = 0
                for ch in pat:
                    k = k * 4 + STATE[ch]
                dense[k] = v
            out[f"{tag}_parent"] = parent; out[f"{tag}_leaf"] = leaf; out[f"{tag}_trans"] = trans
            out[f"{tag}_probs"] = dense; out[f"{tag}_n"] = n
        # (b) banned-pattern flattenings (constructions.py:94-105) and their caller flattening_rank_k_approximation
        # (phylogenetics.py:343-361): a 7-pattern 4-taxon table, every letter on either side; one 10-taxon split
        t7 = {"ATCG": 2 / 5, "GATC": 1 / 5, "CGAT": 1 / 5, "TCGA": 1 / 5, "AATT": 0.1, "AACC": 0.05, "ACAC": 0.05}
        k7, p7 = pack(t7)
        out["t7_keys"] = k7; out["t7_probs"] = p7
        taxa4 = ["0", "1", "2", "3"]
        split4 = (("0", "1"), ("2", "3"))
        for ch in "ACGT":
            for side in ("row", "col"):
                m = sfb(split4, t7, taxa4, **{f"ban_{side}_patterns": ch}).tocoo()
                out[f"t7_ban_{side}_{ch}_r"] = m.row; out[f"t7_ban_{side}_{ch}_c"] = m.col; out[f"t7_ban_{side}_{ch}_v"] = m.data
        out["t7_rank_k"] = np.asarray(flattening_rank_k_approximation(split4, t7).todense())
        g = np.load(os.path.join(OUT, "n10_L10k.npz"))
        names10 = [str(np.base_repr(i, base=max(i + 1, 2))) for i in range(10)]
        table10 = {}
        for k, v in zip(g["keys"].tolist(), g["probs"].tolist()):
            table10["".join("ACGT"[(k >> (2 * (9 - t))) & 3] for t in range(10))] = v
        sid = 7
        m10 = int(g["masks"][sid])
        split10 = (tuple(names10[t] for t in range(10) if (m10 >> t) & 1), tuple(names10[t] for t in range(10) if not (m10 >> t) & 1))
        out["n10_split_id"] = sid
        for ch, side in (("A", "row"), ("T", "col"), ("G", "row")):
            m = sfb(split10, table10, names10, **{f"ban_{side}_patterns": ch}).tocoo()
            order = np.lexsort((m.col, m.row))
            out[f"n10_ban_{side}_{ch}_r"] = m.row[order]; out[f"n10_ban_{side}_{ch}_c"] = m.col[order]
            out[f"n10_ban_{side}_{ch}_v"] = m.data[order]
        rk = flattening_rank_k_approximation(split10, table10).tocoo()
        order = np.lexsort((rk.col, rk.row))
        pick = order[:: max(1, len(order) // 300)]
        out["n10_rank_k_shape"] = np.array(rk.shape); out["n10_rank_k_nnz"] = rk.nnz
        out["n10_rank_k_sum"] = rk.data.sum(); out["n10_rank_k_fro"] = np.sqrt((rk.data ** 2).sum())
        out["n10_rank_k_r"] = rk.row[pick]; out["n10_rank_k_c"] = rk.col[pick]; out["n10_rank_k_v"] = rk.data[pick]
        # (c) frobenius_norm (matrix.py:7-14), all three branches, on the same 10-taxon split
        import pandas as pd
        Fs = flattening(split10, table10)
        Fr = flattening(split10, table10, FlatFormat.reduced)
        df = pd.DataFrame({"pattern": list(table10.keys()), "prob": list(table10.values())})
        out["fro_sparse"] = frobenius_norm(Fs); out["fro_dense"] = frobenius_norm(Fr)
        out["fro_table"] = frobenius_norm(None, data_table=df)
        # (d) wall time of the reference vs the oracle's loops layer on config 1 (README: first 100 splits, reduced)
        splits = list(splitp.all_splits(splitp.trees.balanced_newick_tree(10, 0.05)))[:100]
        t0 = time.time()
        ref_scores = [split_score(flattening(s, table10, FlatFormat.reduced)) for s in splits]
        t_ref = time.time() - t0
        t0 = time.time()
        own_scores = [O.split_score(O.flattening(s, table10, "reduced")) for s in splits]
        t_own = time.time() - t0
        assert np.array_equal(np.array(ref_scores), np.array(own_scores))
        out["time_ref_100"] = t_ref; out["time_oracle_100"] = t_own; out["time_cpus"] = os.cpu_count()
        # (e) searched and NOT found: a matrix of rank <= 4 on which the reference's unclamped dense path returns nan
        # (phylogenetics.py:293-300).  Its denominator adds non-negative terms to the very same partial sum that forms
        # the numerator, so the ratio cannot exceed 1; 3000 random low-rank matrices (integer and real factors) all gave
        # 0.0 or ~1e-8.  nan only arises from the all-zero matrix (0/0), which `degenerate` and the GPU tests cover.
        z = np.zeros((6, 9))
        with np.errstate(invalid="ignore", divide="ignore"):
            out["zero_matrix_score"] = split_score(z)
        np.savez_compressed(os.path.join(OUT, "extras.npz"), **out)
        print(f"extras done: reference {t_ref:.2f} s, oracle {t_own:.2f} s for 100 splits (ratio {t_own / t_ref:.2f})")

    # --------------------------------------------------------- degenerate cases
    if want("degenerate"):
        out = dict(versions=versions)
        rng = np.random.default_rng(5)
        # rank<=4 dense matrix: reference dense path gives ~0 (or nan when rounding goes negative)
        A = rng.integers(0, 50, size=(12, 4)).astype(float); B = rng.integers(0, 50, size=(4, 30)).astype(float)
        M = A @ B
        out["rank4"] = M
        with np.errstate(invalid="ignore"):
            out["rank4_score"] = split_score(M)
        M2 = rng.integers(0, 20, size=(3, 40)).astype(float)      # min(shape) < 4
        out["thin3"] = M2; out["thin3_score"] = split_score(M2)
        M3 = rng.integers(0, 9, size=(40, 25)).astype(float)      # generic full rank
        out["generic"] = M3; out["generic_score"] = split_score(M3)
        M4 = rng.random((33, 70))                                 # non-integer entries
        out["realvalued"] = M4; out["realvalued_score"] = split_score(M4)
        # sparse path on a generic sparse matrix
        from scipy.sparse import dok_matrix
        S = dok_matrix((64, 256))
        for _ in range(600):
            S[rng.integers(0, 64), rng.integers(0, 256)] = float(rng.integers(1, 30))
        coo = S.tocoo()
        out["sp_rows"] = coo.row; out["sp_cols"] = coo.col; out["sp_vals"] = coo.data; out["sp_shape"] = np.array(S.shape)
        out["sp_score"] = split_score(S)
        np.savez_compressed(os.path.join(OUT, "degenerate.npz"), **out)
        print("degenerate done")
    return 0


if __name__ == "__main__":
    sys.exit(main())
