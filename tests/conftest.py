import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

    return load


def taxa_names(n):
    return [np.base_repr(i, base=max(i + 1, 2)) for i in range(n)]


def mask_to_split(mask, n, names=None):
    """left-side bitmask (bit t = taxon t) -> (left, right) tuples of taxon names, index order."""
    names = names or taxa_names(n)
    left = tuple(names[t] for t in range(n) if (mask >> t) & 1)
    right = tuple(names[t] for t in range(n) if not (mask >> t) & 1)
    return left, right
