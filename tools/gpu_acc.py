"""Accuracy / iteration statistics of the sparse route against the golden scores (GPU box)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splitp_amd as sp
from oracle import splitp_oracle as O
from tests.conftest import mask_to_split, taxa_names
for name in ("n10_L100k", "n10_L10k"):
    g = np.load(f"tests/golden/{name}.npz")
    names = taxa_names(10)
    splits = [mask_to_split(int(m), 10, names) for m in g["masks"]]
    dev = sp.DeviceAlignment.from_table(O.unpack_table(g["keys"], g["probs"], 10), taxa=names)
    ss, sts = sp.score_splits(dev, splits, route="sparse", return_status=True)
    err = np.abs(ss - g["scores"])
    its = sts >> 8
    print(name, "max err %.2e" % err.max(), "its histogram", dict(zip(*np.unique(its, return_counts=True))), "flags", np.unique(sts & 3))
    worst = np.argsort(-err)[:5]
    for i in worst:
        print("   split", i, "k", min(len(splits[i][0]), len(splits[i][1])), "err %.2e" % err[i], "its", its[i], "score", ss[i])
