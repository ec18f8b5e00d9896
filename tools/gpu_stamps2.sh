set -e
cd $GRAFT_REPO_ROOT
cd splitp_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DSPK_STAMPS -c sparse.hip -o /tmp/sparse_st.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsplitp_hip.so api.o flatten.o gram.o gram_i8.o eigen.o /tmp/sparse_st.o subflat.o hist.o
cd ../..
python - <<'PY'
import sys, ctypes as C, numpy as np
sys.path.insert(0,'.')
import splitp_amd as sp
from splitp_amd import synthetic as syn, batch, _lib
n, L = 10, 100_000
names = syn.taxa_names(n)
sites = syn.simulate_sites(n, L, 0.05, seed=1)
keys, counts = syn.pattern_table(sites)
dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=L, taxa=names)
splits = list(sp.all_splits(names))
lib = dev.ctx._lib
lib.sp_debug_spk_stamps.argtypes = [C.POINTER(C.c_longlong)]
for k in (5, 4, 3):
    sub = [s for s in splits if min(len(s[0]), len(s[1])) == k]
    taxa_arr, a_arr = batch.encode_splits(sub, dev, n)
    for rep in range(2):
        batch.score_encoded(dev, taxa_arr, a_arr, 3)
    out = (C.c_longlong * 64)()
    lib.sp_debug_spk_stamps(out)
    st = np.array(out[:12], dtype=np.int64); d = np.diff(st)
    names_ = ["bitmaps+trace", "CSC build", "CSR build", "(carve)", "diag/G", "start block", "ritz_orth(init)", "(loop entry)", "spmm1", "spmm2", "ritz_orth(it1)", "rest of iterations"]
    names2 = ["stage+bitmaps", "CSC build", "CSR build", "start(argmax)", "V init/G", "ritz(init)", "-", "spmm1", "spmm2", "ritz(it1)", "rest"]
    print(f"k={k}:", "  ".join(f"{a}={b}" for a, b in zip(names2, d)), " total", st[11]-st[0], " | spmm1 light", out[20]-st[7], "heavy", st[8]-out[20], " spmm2 light", out[21]-st[8], "heavy", st[9]-out[21], "nheavy?", out[30], "| ritz(it2): gram", out[41]-out[40], "4x4", out[42]-out[41], "apply", out[43]-out[42], "polish", out[44]-out[43])
PY
