"""Synthetic Jukes-Cantor alignments on balanced trees (benchmark / test input generator).

This is NOT part of the drop-in path; it replaces, for benchmarking only, the reference's
per-site Python simulator (splitp/simulation.py:9-56, 73 s per 100 k sites) with a vectorised
NumPy walk down the same tree shape:

  * topology of splitp/trees.py:6-29 `balanced_newick_tree(n, t)` (n even; n = 10 gives
    ((((0,1),2),(3,4)),((5,6),(7,(8,9))))), every one of the 2n-2 branches of length t;
  * Jukes-Cantor: P(stay) = 1/4 + 3/4 exp(-4t/3) per branch, the other three states equally
    likely (the rate argument of GTR.JukesCantor cancels, splitp/model.py:58-61);
  * uniform root state (splitp/simulation.py:28).

It is statistically equivalent to, but not stream-identical with, the reference simulator
(different RNG); the CPU checker and the HIP path are always fed the same generated table.
"""
from __future__ import annotations

from math import exp, floor

import numpy as np

STATES = "ACGT"


def balanced_tree(n_taxa: int):
    """Nested-tuple topology of the reference's balanced tree; leaves are ints 0..n-1 in
    left-to-right order."""
    if n_taxa % 2 != 0 or n_taxa < 2:
        raise ValueError("balanced trees need an even number of taxa")
    counter = [0]

    def leaf():
        counter[0] += 1
        return counter[0] - 1

    def sub(nt, left):
        nt = int(nt)
        if nt == 1:
            return leaf()
        if nt == 2:
            return (leaf(), leaf())
        if nt == 3:
            if left:
                return ((leaf(), leaf()), leaf())
            return (leaf(), (leaf(), leaf()))
        if nt % 2 == 0:
            return (sub(nt // 2, True), sub(nt // 2, False))
        return (sub(floor(nt / 2) + int(left), True), sub(floor(nt / 2) + int(not left), False))

    if n_taxa == 2:
        return (leaf(), leaf())
    return (sub(n_taxa // 2, True), sub(n_taxa // 2, False))


def tree_splits(tree, n_taxa):
    """Non-trivial splits (as frozensets of the side containing fewer taxa / not taxon 0)
    displayed by the tree."""
    out = []

    def walk(node):
        if isinstance(node, int):
            return {node}
        s = set()
        for ch in node:
            s |= walk(ch)
        if 2 <= len(s) <= n_taxa - 2:
            out.append(frozenset(s))
        return s

    walk(tree)
    return set(out)


def simulate_sites(n_taxa: int, n_sites: int, branch_length: float = 0.05, seed: int = 1) -> np.ndarray:
    """(n_sites, n_taxa) uint8 digits 0..3 (A,C,G,T)."""
    rng = np.random.default_rng(seed)
    tree = balanced_tree(n_taxa)
    p_change = 1.0 - (0.25 + 0.75 * exp(-4.0 * branch_length / 3.0))
    out = np.empty((n_sites, n_taxa), dtype=np.uint8)

    def mutate(state):
        change = rng.random(n_sites) < p_change
        shift = rng.integers(1, 4, size=n_sites, dtype=np.uint8)
        return np.where(change, (state + shift) & 3, state).astype(np.uint8)

    def walk(node, parent_state):
        state = mutate(parent_state)
        if isinstance(node, int):
            out[:, node] = state
        else:
            for ch in node:
                walk(ch, state)

    root = rng.integers(0, 4, size=n_sites, dtype=np.uint8)
    for ch in tree:
        walk(ch, root)
    return out


def site_keys(sites: np.ndarray) -> np.ndarray:
    """Pack (L, n) digits into uint64 keys, taxon 0 most significant."""
    n = sites.shape[1]
    keys = np.zeros(sites.shape[0], dtype=np.uint64)
    for t in range(n):
        keys = (keys << np.uint64(2)) | sites[:, t].astype(np.uint64)
    return keys


def pattern_table(sites: np.ndarray):
    """(keys sorted ascending = the A<C<G<T pattern order of simulation.py:51-54, counts int64)."""
    keys = site_keys(sites)
    uk, cnt = np.unique(keys, return_counts=True)
    return uk, cnt.astype(np.int64)


def table_as_dict(keys, counts, n_taxa, total=None):
    """The dict the reference's generate_alignment returns: pattern string -> count/float(L)."""
    total = float(counts.sum() if total is None else total)
    out = {}
    for k, c in zip(keys.tolist(), counts.tolist()):
        out["".join(STATES[(k >> (2 * (n_taxa - 1 - t))) & 3] for t in range(n_taxa))] = c / total
    return out


def sequences_ascii(sites: np.ndarray) -> np.ndarray:
    """(n_taxa, L) uint8 ASCII rows (FASTA-like sequences)."""
    lut = np.frombuffer(STATES.encode(), dtype=np.uint8)
    return np.ascontiguousarray(lut[sites].T)


def taxa_names(n_taxa):
    """Leaf names of trees.py:15-16 ('0'-'9','A'-'Z' for n <= 36)."""
    if n_taxa <= 36:
        return [np.base_repr(i, base=max(i + 1, 2)) for i in range(n_taxa)]
    return [f"t{i}" for i in range(n_taxa)]
