# Phase stamps of k_subscore_pair (diagnostic build -DSUBP_STAMPS): cycles of one wave per phase of one of its longest pairs,
# 16 and 20 taxa, all splits (the wave shares its SIMD with 3 others: wall cycles, not issue slots):  bash tools/gpu_subpair_stamps.sh
set -e
cd $GRAFT_REPO_ROOT
bash tools/variant_lib.sh subflat_pair.hip /tmp/lib_subp_stamps.so -DSUBP_STAMPS $SUBP_EXTRA
SPLITP_LIB=/tmp/lib_subp_stamps.so python - <<'PY'
import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import splitp_amd as sp
from splitp_amd import synthetic as syn, simulation as sim, _lib
import torch
for n in (16, 20):
    dev = sim.generate_device_alignment(syn.balanced_tree(n), sim.JukesCantor(), 1_000_000, seed=5, branch_length=0.05)
    lib = dev.ctx._lib
    lib.sp_debug_subp_stamps.argtypes = [C.POINTER(C.c_longlong)]
    n_got = C.c_int64()
    sc = torch.zeros(1 << 19, dtype=torch.float64, device="cuda")
    st = torch.zeros(1 << 19, dtype=torch.int32, device="cuda")
    for rep in range(2):
        _lib.check(lib.sp_score_all_splits_shard(dev.handle, _lib.SP_METHOD_SUBFLATTENING, 0, 0, 0, 1, C.byref(n_got), None,
                                                 C.c_void_p(sc.data_ptr()), None, C.c_void_p(st.data_ptr())))
        torch.cuda.synchronize()
    out = (C.c_longlong * 16)()
    lib.sp_debug_subp_stamps(out)
    g = np.array(out[6:12], dtype=np.int64)
    print(f"   Gram phase of the pair, split A: index tables (global loads of the split) {g[0]-out[0]}  products + stores {g[1]-g[0]}  row loads {g[2]-g[1]};"
          f"  split B: {g[3]-g[2]} / {g[4]-g[3]} / {g[5]-g[4]}")
    o = np.array(out[:6], dtype=np.int64)
    d = np.diff(o)
    print(f"{n} taxa, {n_got.value} splits: two Gram matrices + rows into registers {d[0]}  tridiagonalisation {d[1]}  table + scaling {d[2]}  Sturm passes {d[3]}  score {d[4]}  total {o[5]-o[0]} cycles of one pair (s_memtime)")
PY
