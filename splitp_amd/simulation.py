"""Alignment simulator on the device.

Mirror of reference splitp/simulation.py:9-56: `generate_alignment(tree, model, sequence_length)` walks every site
independently down the tree - uniform root state (:28), at each node a new state drawn from column `state` of the
branch's transition matrix (:17-18) - counts the site patterns and returns `pattern -> count / sequence_length`
sorted by pattern (:43-56).  Here the walk is one HIP thread per site and the counting is the site-pattern
histogram kernel, so the table is born on the device (`generate_device_alignment`); the reference's dict is built
from it only when asked for (`generate_alignment`).  73 s per 100 k sites on the reference's CPU path.

The random streams differ from the reference's (`random.choices`), so outputs are statistically - not bit-wise -
equal; tests compare against the exact pattern distribution (oracle) and check the sampling semantics with
deterministic transition matrices.

Trees: (a) the reference's `Phylogeny` (anything with `.networkx_graph` - a rooted DiGraph whose nodes carry
`branch_length` and / or `transition_matrix` - and `.taxa`), (b) nested tuples of taxon indices as
`splitp_amd.synthetic.balanced_tree` builds them, every branch of length `branch_length`.
Models: anything with `transition_matrix(t)` returning the 4 x 4 matrix of a branch of length t (the reference's
`model.GTR...` objects qualify), or `JukesCantor()` below (closed form), or None to take the matrices stored on the
tree's nodes (simulation.py:13-14)."""
from __future__ import annotations

import ctypes as C
from math import exp

import numpy as np

from . import _lib
from .device import DeviceAlignment, get_context


class JukesCantor:
    """Jukes-Cantor transition matrices in closed form: the reference's GTR.JukesCantor (model.py:66-75) normalises
    its rate matrix to one expected substitution per unit time, so `rate` cancels and
    P(t) = 1/4 + 3/4 exp(-4t/3) on the diagonal, 1/4 - 1/4 exp(-4t/3) elsewhere (= expm(t Q), model.py:14-16)."""

    name = "Jukes-Cantor model"

    def __init__(self, rate=1):
        self.rate = rate

    def transition_matrix(self, t):
        e = exp(-4.0 * t / 3.0)
        m = np.full((4, 4), 0.25 - 0.25 * e)
        np.fill_diagonal(m, 0.25 + 0.75 * e)
        return m


def _arrays_from_networkx(tree, model):
    g = tree.networkx_graph
    taxa = list(tree.taxa)
    roots = [n for n, d in g.in_degree() if d == 0]
    if len(roots) != 1:
        raise ValueError("the tree needs exactly one root")
    order, parent = [roots[0]], [-1]
    i = 0
    while i < len(order):          # parents first (breadth first)
        for child in g.successors(order[i]):
            order.append(child)
            parent.append(i)
        i += 1
    leaf, mats = [], []
    for k, node in enumerate(order):
        is_leaf = g.out_degree(node) == 0
        leaf.append(taxa.index(str(node)) if is_leaf and str(node) in taxa else (taxa.index(node) if is_leaf else -1))
        if k == 0:
            mats.append(np.eye(4))
        elif model is None:
            mats.append(np.asarray(g.nodes[node]["transition_matrix"], dtype=np.float64))
        else:
            mats.append(np.asarray(model.transition_matrix(g.nodes[node]["branch_length"]), dtype=np.float64))
    return parent, leaf, mats, taxa


def _arrays_from_nested(tree, model, branch_length):
    if model is None:
        raise ValueError("a nested-tuple tree needs a model (it carries no transition matrices)")
    m = np.asarray(model.transition_matrix(branch_length), dtype=np.float64)
    parent, leaf, mats = [-1], [-1], [np.eye(4)]

    def walk(node, par):
        idx = len(parent)
        parent.append(par)
        mats.append(m)
        if isinstance(node, (int, np.integer)):
            leaf.append(int(node))
        else:
            leaf.append(-1)
            for ch in node:
                walk(ch, idx)

    for ch in tree:
        walk(ch, 0)
    n = max(leaf) + 1
    return parent, leaf, mats, [np.base_repr(i, base=max(i + 1, 2)) if n <= 36 else f"t{i}" for i in range(n)]


def tree_arrays(tree, model=None, branch_length=None):
    """(parent, leaf_taxon, transition[n_nodes, 4, 4], taxa) in the parents-first form sp_simulate_alignment takes."""
    if hasattr(tree, "networkx_graph"):
        parent, leaf, mats, taxa = _arrays_from_networkx(tree, model)
    else:
        if branch_length is None:
            raise ValueError("branch_length is required for a nested-tuple tree")
        parent, leaf, mats, taxa = _arrays_from_nested(tree, model, branch_length)
    trans = np.ascontiguousarray(np.stack(mats), dtype=np.float64)
    if trans.shape[1:] != (4, 4):
        raise ValueError("transition matrices must be 4 x 4 (DNA state space A, C, G, T)")
    return np.asarray(parent, dtype=np.int32), np.asarray(leaf, dtype=np.int32), trans, taxa


def generate_device_alignment(tree, model, sequence_length, seed=None, branch_length=None, device=None):
    """Simulate `sequence_length` sites and leave the pattern table on the device (a DeviceAlignment)."""
    parent, leaf, trans, taxa = tree_arrays(tree, model, branch_length)
    if seed is None:
        seed = int(np.random.SeedSequence().generate_state(1, dtype=np.uint64)[0])
    ctx = get_context(device)
    h = C.c_void_p()
    _lib.check(ctx._lib.sp_simulate_alignment(ctx.handle, len(parent), _lib._ptr(parent, C.c_int32),
                                              _lib._ptr(leaf, C.c_int32), _lib._ptr(trans, C.c_double), len(taxa),
                                              int(sequence_length), C.c_uint64(int(seed) & (2 ** 64 - 1)), C.byref(h)))
    return DeviceAlignment(h, ctx, len(taxa), taxa)


def generate_alignment(tree, model, sequence_length, seed=None, branch_length=None):
    """reference: splitp/simulation.py:43-56 - dict pattern -> count / float(sequence_length), patterns in A < C < G < T
    order.  (`seed` and `branch_length` are extensions; see the module docstring.)"""
    dev = generate_device_alignment(tree, model, sequence_length, seed=seed, branch_length=branch_length)
    keys, weights, _ = dev.fetch()
    n = dev.n_taxa
    out = {}
    for k, w in zip(keys.tolist(), weights.tolist()):
        out["".join("ACGT"[(k >> (2 * (n - 1 - t))) & 3] for t in range(n))] = w
    return out
