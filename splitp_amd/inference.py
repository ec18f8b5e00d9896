"""erickson_SVD - the in-tree caller of the hot path (reference splitp/phylogenetics.py:99-171), re-expressed on the
batched device API: every agglomeration round scores all its not-yet-seen candidate splits in ONE `score_splits`
call with the alignment resident in HBM (the reference scores them one by one through flattening() + split_score()).

Same contract as the reference: returns the list of chosen splits (each a sorted tuple of two sorted taxon tuples),
memoises scores across rounds (:122-124, :142), takes the first minimum in pair-enumeration order (:146), merges the
winning pair (:160-170).  Supported methods: Method.flattening, Method.subflattening and Method.mutual_information
(the reference's third branch, :135-140: flattening_rank_1_approximation_divergence, batched in csrc/divergence.hip)."""
from itertools import combinations

import numpy as np

from .batch import score_splits
from .device import as_device_alignment
from .enums import Method


def _flatten(group):
    return tuple(group) if not isinstance(group, str) else (group,)


def erickson_SVD(alignment, taxa=None, method=Method.flattening, show_work=False):
    name = getattr(method, "name", method)
    if name not in ("flattening", "subflattening", "mutual_information"):
        raise NotImplementedError(f"erickson_SVD on the device supports flattening / subflattening / mutual_information, "
                                  f"not {method!r}")
    first = next(iter(alignment.keys())) if not hasattr(alignment, "n_taxa") else None
    num_taxa = alignment.n_taxa if first is None else len(first)
    if taxa is None:  # reference :148-152
        taxa = [str(np.base_repr(i, base=max(i + 1, 2))) if num_taxa <= 36 else f"t{i}" for i in range(num_taxa)]
    leaf_order = [t for g in taxa for t in _flatten(g)]

    dev = as_device_alignment(alignment)
    if getattr(dev, "taxa", None) is None:
        dev.taxa = tuple(sorted(leaf_order))  # the reference resolves taxa as sorted(union of the split halves)
    all_scores = {}
    true_splits = []
    while len(true_splits) < num_taxa - 2:
        current = list(taxa)
        pairs, splits = [], []
        for pair in combinations(current, 2):
            flat_pair = tuple(sorted(e for tup in pair for e in _flatten(tup)))
            other = tuple(sorted(e for tup in current for e in _flatten(tup) if e not in flat_pair))
            pairs.append(pair)
            splits.append((flat_pair, other))
        todo = [s for s in dict.fromkeys(splits) if s not in all_scores]
        scorable = [s for s in todo if len(s[0]) >= 1 and len(s[1]) >= 1]
        if scorable:
            vals = score_splits(dev, scorable, method=Method[name])
            for s, v in zip(scorable, vals):
                all_scores[s] = float(v)
        for s in todo:
            all_scores.setdefault(s, float("inf"))  # a split with an empty side cannot be scored (reference: np.inf)
        scored = [(pair, split, all_scores[split]) for pair, split in zip(pairs, splits)]
        if show_work:
            print(f"Scores: { {p: (p, s, v) for p, s, v in scored} }")
        best_pair, best_split, best_score = min(scored, key=lambda x: x[2])
        true_splits.append(tuple(sorted(best_split)))
        taxa = tuple([e for e in taxa if (e not in best_split[0] and not set(e).issubset(best_split[0]))]
                     + [best_split[0]])
    return true_splits
