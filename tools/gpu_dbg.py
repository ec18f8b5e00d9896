import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splitp_amd as sp
from oracle import splitp_oracle as O
from tests.conftest import mask_to_split, taxa_names
g = np.load("tests/golden/n10_L10k.npz")
names = taxa_names(10)
splits = [mask_to_split(int(m), 10, names) for m in g["masks"]]
dev = sp.DeviceAlignment.from_table(O.unpack_table(g["keys"], g["probs"], 10), taxa=names)
print(dev.info())
sd, std = sp.score_splits(dev, splits, route="dense", return_status=True)
ss, sts = sp.score_splits(dev, splits, route="sparse", return_status=True)
err = np.abs(ss - g["scores"])
bad = np.where(err > 1e-10)[0]
print("dense max err", np.abs(sd - g["scores"]).max(), "sparse bad count", len(bad))
for i in bad[:20]:
    print(i, "k=", min(len(splits[i][0]), len(splits[i][1])), "shape", g["shapes"][i], "ref", g["scores"][i], "sparse", ss[i], "its", sts[i] >> 8, "flags", sts[i] & 3)
