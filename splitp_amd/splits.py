"""Host-side split enumeration in the reference's order (splitp/splits.py:27-59).

The order matters: multi-GPU sharding deals splits to ranks by their position in this
sequence, so every rank must enumerate identically."""
from itertools import combinations
from math import floor


def all_splits(tree_or_taxa, trivial=False, size=None, randomise=False, string_format=False):
    """Same generator contract as reference splits.all_splits, accepting either an object
    with `.taxa` (the reference's Phylogeny) or a plain sequence of taxon names."""
    taxa = list(getattr(tree_or_taxa, "taxa", tree_or_taxa))
    n = len(taxa)
    if string_format and n > 35:
        raise ValueError("Cannot generate splits for more than 35 taxa in string format. Use string_format=False.")
    sizes = [size] if size is not None else list(range(1 if trivial else 2, floor(n / 2) + 1))
    for bal in sizes:
        even = bal == n / 2
        combos = combinations(taxa[1:], bal - 1) if even else combinations(taxa, bal)
        if randomise:
            from numpy.random import shuffle

            combos = list(combos)
            shuffle(combos)
        for left in combos:
            if even:
                left = (taxa[0],) + tuple(left)
            right = tuple(sorted(set(taxa) - set(left), key=taxa.index))
            left = tuple(sorted(left, key=taxa.index))
            if taxa[0] in right:
                left, right = right, left
            if string_format:
                if not all(len(str(t)) == 1 for t in taxa):
                    raise ValueError("Cannot produce string format for split with taxa name of length > 1.")
                yield f'{"".join(left)}|{"".join(right)}'
            else:
                yield (left, right)
