set -e
cd $GRAFT_REPO_ROOT
cd splitp_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DEIG_STAMPS -c eigen.hip -o /tmp/eigen_st.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsplitp_hip.so api.o flatten.o gram.o /tmp/eigen_st.o subflat.o hist.o
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
sub = [s for s in splits if min(len(s[0]), len(s[1])) == 5] + splits[:400]
taxa_arr, a_arr = batch.encode_splits(sub, dev, n)
for rep in range(3):
    batch.score_encoded(dev, taxa_arr, a_arr, 0)
lib = dev.ctx._lib
out = (C.c_longlong * 64)()
lib.sp_debug_eig_stamps.argtypes = [C.POINTER(C.c_longlong)]
print("rc", lib.sp_debug_eig_stamps(out))
st = np.array(out[:13], dtype=np.int64)
d = np.diff(st)
print("Y->LDS", d[0], "ritz_orth16", d[1], "jacobi sweeps", out[20], "NS iters", out[21])
PY
