cd $GRAFT_REPO_ROOT
for cap in 0 30000 12000; do
  for t in a b; do SPLITP_DEBUG_LDS_CAP=$cap python tools/gpu_determinism_xproc.py 31011 400 12 /tmp/xp_${cap}_$t.npy 2>&1 | grep -v amdgpu.ids | tail -2; done
  python - <<PY
import numpy as np
a, b = np.load("/tmp/xp_${cap}_a.npy"), np.load("/tmp/xp_${cap}_b.npy")
print("lds_cap $cap: cross-process determinism: %d scores, %d differ bitwise" % (a.size, int((a.view(np.uint64) != b.view(np.uint64)).sum())))
PY
done
python - <<'PY'
# 12-taxon 100 k-site table (the slow kernel's forms at their real sizes): the same call in this process three times
import sys, numpy as np
sys.path.insert(0, '.')
import splitp_amd as sp
from splitp_amd import synthetic as syn, simulation as sim
n = 12
names = syn.taxa_names(n)
outs = []
for rep in range(3):
    dev = sim.generate_device_alignment(syn.balanced_tree(n), sim.JukesCantor(), 100_000, seed=77, branch_length=0.05)
    dev.taxa = tuple(names)
    outs.append(sp.score_all_splits(dev))
print("12 taxa x 100 k sites, 2035 splits, three fresh alignments: bitwise differences", int((outs[0].view(np.uint64) != outs[1].view(np.uint64)).sum()), int((outs[0].view(np.uint64) != outs[2].view(np.uint64)).sum()))
PY
