"""splitp_amd - MI355X-native implementation of SplitP's flattening / subflattening /
split_score hot path (reference: js51/SplitP v0.3.2, splitp/constructions.py,
splitp/phylogenetics.py:280-328, splitp/matrix.py).

Drop-in surface (same names / signatures as `import splitp`, reference splitp/__init__.py:15-17):
    flattening, subflattening, split_score, FlatFormat, all_splits
plus the device-resident batched form of the README loop:
    DeviceAlignment, score_splits
Host code is Python; all compute is hand-written HIP for gfx950 behind a ctypes C ABI
(include/splitp_hip.h).  There is no CPU fallback."""
from . import constants, constructions, enums, matrix, phylogenetics, simulation, splits  # noqa: F401
from .batch import NodeScorer, encode_all_splits, score_all_splits, score_splits  # noqa: F401
from .constructions import (flattening, sparse_flattening_with_banned_patterns,  # noqa: F401
                            subflattening)
from .device import DeviceAlignment, get_context  # noqa: F401
from .enums import FlatFormat, Method  # noqa: F401
from .inference import erickson_SVD  # noqa: F401
from .matrix import frobenius_norm, is_sparse  # noqa: F401
from .phylogenetics import split_score  # noqa: F401
from .simulation import generate_alignment  # noqa: F401
from .splits import all_splits  # noqa: F401
from ._lib import SplitPDeviceError  # noqa: F401

__version__ = "0.4.0"


def install_as_splitp(force=False):
    """Make `import splitp` (and splitp.constructions / .phylogenetics / .matrix / .enums / .constants / .splits) resolve to
    this package: the literal drop-in for code written against the reference (splitp/__init__.py:15-18 names the surface).
    Refuses to shadow an already imported reference package unless force=True.  Returns the module registered."""
    import sys

    me = sys.modules[__name__]
    have = sys.modules.get("splitp")
    if have is not None and have is not me and not force:
        raise ImportError("a module named 'splitp' is already imported; pass force=True to replace it")
    sys.modules["splitp"] = me
    for sub in ("constructions", "phylogenetics", "matrix", "enums", "constants", "splits", "simulation"):
        sys.modules["splitp." + sub] = sys.modules[__name__ + "." + sub]
    return me
