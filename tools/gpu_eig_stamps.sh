# cycle stamps of k_eig_rr's workgroup 0 (the heaviest split) in the dense route (GPU box): bash tools/gpu_eig_stamps.sh
set -e
cd $GRAFT_REPO_ROOT
for round in ${EIG_ROUNDS:-0 1 3}; do
bash tools/variant_lib.sh eigen.hip /tmp/lib_eig_st.so -DEIG_STAMPS $EIG_EXTRA -DEIG_STAMP_ROUND=$round 2>/dev/null
(SPLITP_LIB=/tmp/lib_eig_st.so python - <<PY
import sys, ctypes as C, numpy as np
sys.path.insert(0,'.')
import splitp_amd as sp
from splitp_amd import synthetic as syn, batch, _lib
n, L = 10, 100_000
names = syn.taxa_names(n)
keys, counts = syn.pattern_table(syn.simulate_sites(n, L, 0.05, seed=1))
dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=L, taxa=names)
taxa_arr, a_arr = sp.encode_all_splits(n)
for rep in range(3):
    sc, st = batch.score_encoded(dev, taxa_arr, a_arr, _lib.SP_METHOD_FLATTENING_DENSE)
lib = dev.ctx._lib
lib.sp_debug_eig_stamps.argtypes = [C.POINTER(C.c_longlong)]
out = (C.c_longlong * 64)()
lib.sp_debug_eig_stamps(out)
o = np.array(out[:], dtype=np.int64)
d = lambda a, b: int(o[a] - o[b])
print("round $round (cycles): load_Y=%d proj=%d gram=%d jacobi=%d T+rowmul=%d polish0=%d polish1=%d polish2=%d ritz_total=%d end(3)=%d accept(5)=%d" % (
    d(1,0), d(4,1), d(10,4), d(11,10), d(12,11), d(13,12), d(14,13), d(15,14), d(2,4), d(3,2), d(5,2)))
if "$EIG_CERT":
    lib.sp_debug_eig_dump.argtypes = [C.POINTER(C.c_double)]
    dd = (C.c_double * 256)()
    lib.sp_debug_eig_dump(dd)
    v = np.array(dd[:])
    dl, rho = v[:126], v[128:254]
    bound = dl * rho**2 / (1 - rho**2)
    np.set_printoptions(linewidth=200, precision=2)
    print("   certified stop at product 3, first 126 workgroups (long sides): delta/s", np.sort(dl)[[0, 31, 63, 94, 125]], "rho", np.sort(rho)[[0, 31, 63, 94, 125]])
    print("   bound delta q/(1-q) quantiles", np.sort(bound)[[0, 31, 63, 94, 125]], " passes (<= 1e-15):", int((bound <= 1e-15).sum()), " with q/30:", int((bound / 30 <= 1e-15).sum()), " with q/100:", int((bound / 100 <= 1e-15).sum()))
if "$EIG_DUMP":
    lib.sp_debug_eig_dump.argtypes = [C.POINTER(C.c_double)]
    dd = (C.c_double * 256)()
    lib.sp_debug_eig_dump(dd)
    H = np.array(dd[:]).reshape(16, 16)
    np.set_printoptions(linewidth=250, precision=1)
    print("   projected matrix before the first Jacobi, log10 |h_ij| / max diag (first 10 rows):")
    print(np.round(np.log10(np.abs(H) / np.abs(np.diag(H)).max() + 1e-99), 1)[:10])
print("   first Jacobi per sweep [(-10 log10 max|v|/dmax) * 1000 + open pairs]:", o[24:34].tolist())
print("   jacobi calls %d: last sweep index reached %s, rel*1e30 at that check %s" % (o[39], o[40:40+min(int(o[39]),8)].tolist(), o[48:48+min(int(o[39]),8)].tolist()))
PY
)
done
