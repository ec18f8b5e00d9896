# Phase stamps of k_subscore_tri (diagnostic build -DSUBT_STAMPS): cycles of one wave per phase of one of its longest splits,
# 16 and 20 taxa, all splits (the wave shares its SIMD with 3 others: wall cycles, not issue slots):  bash tools/gpu_subflat_stamps.sh
set -e
cd $GRAFT_REPO_ROOT
bash tools/variant_lib.sh subflat.hip /tmp/lib_subt_stamps.so -DSUBT_STAMPS
SPLITP_LIB=/tmp/lib_subt_stamps.so python - <<'PY'
import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import splitp_amd as sp
from splitp_amd import synthetic as syn, simulation as sim, _lib
import torch
for n in (16, 20):
    dev = sim.generate_device_alignment(syn.balanced_tree(n), sim.JukesCantor(), 1_000_000, seed=5, branch_length=0.05)
    lib = dev.ctx._lib
    lib.sp_debug_subt_stamps.argtypes = [C.POINTER(C.c_longlong)]
    n_got = C.c_int64()
    sc = torch.zeros(1 << 19, dtype=torch.float64, device="cuda")
    st = torch.zeros(1 << 19, dtype=torch.int32, device="cuda")
    for rep in range(2):
        _lib.check(lib.sp_score_all_splits_shard(dev.handle, _lib.SP_METHOD_SUBFLATTENING, 0, 0, 0, 1, C.byref(n_got), None,
                                                 C.c_void_p(sc.data_ptr()), None, C.c_void_p(st.data_ptr())))
        torch.cuda.synchronize()
    out = (C.c_longlong * 16)()
    lib.sp_debug_subt_stamps(out)
    o = np.array(out[:6], dtype=np.int64)
    d = np.diff(o)
    print(f"{n} taxa, {n_got.value} splits: index tables {d[0]}  Gram (MFMA) + store {d[1]}  tridiagonalisation {d[2]}  scaling + 13 Sturm passes {d[3]}  score {d[4]}  total {o[5]-o[0]} cycles (s_memtime)")
PY
