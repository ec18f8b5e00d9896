# A/B of k_sparse_score builds on the GPU box: bash tools/gpu_sparse_variant.sh "<flags A>" "<flags B>" ...
# each build: half-product histogram (tools/gpu_iters.py), max |score - golden|, bench (3 lanes, 1500 steps)
cd $GRAFT_REPO_ROOT
for fl in "$@"; do
  echo "=== flags: $fl"
  bash tools/variant_lib.sh sparse.hip /tmp/lib_v.so $fl || continue
  export SPLITP_LIB=/tmp/lib_v.so
  python tools/gpu_iters.py 2>&1 | grep -v amdgpu.ids
  python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, numpy as np
sys.path.insert(0, '.')
import splitp_amd as sp
from oracle import splitp_oracle as O
from tests.conftest import taxa_names
from tests.test_gpu_parity import mask_to_split
for name in ("n10_L100k", "n10_L10k"):
    g = np.load("tests/golden/%s.npz" % name)
    names = taxa_names(10)
    splits = [mask_to_split(int(m), 10, names) for m in g["masks"]]
    dev = sp.DeviceAlignment.from_table(O.unpack_table(g["keys"], g["probs"], 10), taxa=names)
    s = sp.score_splits(dev, splits)
    print(name, "max |score - reference| = %.2e" % np.abs(s - g["scores"]).max())
PY
  python bench.py --steps 1500 --warmup 50 --no-cpu-baseline --no-pipeline-block 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step %.5f  launch_ms %.5f' % (d['ms_per_step'], d['roofline']['launch_ms']))"
done
