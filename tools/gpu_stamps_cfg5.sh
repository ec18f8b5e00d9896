# cycle stamps of the sparse kernel's slow forms on a 12-taxon 100 k-site table (config 5's shape), one size class per call:
# bash tools/gpu_stamps_cfg5.sh   (GPU box; the stamps are those of the LAST item persistent workgroup 0 scored)
set -e
cd $GRAFT_REPO_ROOT/splitp_amd/csrc
cp ../libsplitp_hip.so /tmp/lib_keep.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $SPK_EXTRA -DSPK_STAMPS -c sparse.hip -o /tmp/sparse_st.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsplitp_hip.so api.o flatten.o gram.o gram_i8.o eigen.o /tmp/sparse_st.o sparse_big.o subflat.o hist.o divergence.o
cd ../..
python - <<'PY'
import sys, ctypes as C, numpy as np
sys.path.insert(0,'.')
import splitp_amd as sp
from splitp_amd import synthetic as syn, batch, _lib
from splitp_amd import simulation as sim
n = 12
names = syn.taxa_names(n)
dev = sim.generate_device_alignment(syn.balanced_tree(n), sim.JukesCantor(), 100_000, seed=101, branch_length=0.05)
dev.taxa = tuple(names)
taxa_arr, a_arr = sp.encode_all_splits(n)
k = np.minimum(a_arr, n - a_arr)
lib = dev.ctx._lib
lib.sp_debug_spk_stamps.argtypes = [C.POINTER(C.c_longlong)]
lib.sp_debug_spk_stamp_block(0)
for kk in (4, 5, 3, 2):
    idx = np.nonzero(k == kk)[0]
    t, a = np.ascontiguousarray(taxa_arr[idx]), np.ascontiguousarray(a_arr[idx])
    for rep in range(2):
        sc, st = batch.score_encoded(dev, t, a, _lib.SP_METHOD_FLATTENING)
    out = (C.c_longlong * 64)()
    lib.sp_debug_spk_stamps(out)
    o = np.array(out[:], dtype=np.int64)
    d = lambda x, y: int(o[x] - o[y])
    print(f"k={kk}: stage={d(1,0)} CSC={d(2,1)} CSR={d(3,2)} start={d(4,3)} init+W1+orth={d(7,4)} spmm_it2={d(9,7)} gram2={d(40,9)} "
          f"chol+orth2={d(41,40)} spmm_it3={d(8,41)} rest={d(11,8)} total={d(11,0)}  half products {np.bincount(st >> 8).nonzero()[0].tolist()}")
PY
cp /tmp/lib_keep.so splitp_amd/libsplitp_hip.so
