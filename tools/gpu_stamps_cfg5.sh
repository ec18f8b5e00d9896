# cycle stamps of the sparse kernel's slow forms on a 12-taxon 100 k-site table (config 5's shape), one size class per call:
# bash tools/gpu_stamps_cfg5.sh   (GPU box; the stamps are those of the LAST item persistent workgroup 0 scored)
set -e
cd $GRAFT_REPO_ROOT
bash tools/variant_lib.sh sparse.hip /tmp/lib_stamps.so $SPK_EXTRA -DSPK_STAMPS
SPLITP_LIB=/tmp/lib_stamps.so python - <<'PY'
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
forms = (C.c_uint * 64)()
FORM = {0: "LDS", 4: "lists-in-global", 12: "lists+W-in-global", 1: "all-global", 3: "all-global wide"}
for kk in (6, 5, 4, 3, 2):
    idx = np.nonzero(k == kk)[0]
    t, a = np.ascontiguousarray(taxa_arr[idx]), np.ascontiguousarray(a_arr[idx])
    sc, st = batch.score_encoded(dev, t, a, _lib.SP_METHOD_FLATTENING)
    lib.sp_debug_spk_forms(forms, 1)
    sc, st = batch.score_encoded(dev, t, a, _lib.SP_METHOD_FLATTENING)
    lib.sp_debug_spk_forms(forms, 1)
    fv = np.array(forms[:]).reshape(16, 4)
    print(f"k={kk}: {len(idx)} items; forms (scored / refused at exit 1, 2, 3):", {FORM.get(f, f): fv[f].tolist() for f in range(16) if fv[f].any()})
    out = (C.c_longlong * 64)()
    lib.sp_debug_spk_stamps(out)
    o = np.array(out[:], dtype=np.int64)
    d = lambda x, y: int(o[x] - o[y])
    if kk >= 5:   # (general path; a Gram-path split sets none of these stamps - round 2's table printed them anyway, as garbage)
        print(f"k={kk} (general path): stage={d(1,0)} CSC list={d(2,1)} CSR list={d(3,2)} start rows={d(4,3)} W init + first half product + orth={d(7,4)} spmm_it2={d(9,7)} gram2={d(40,9)} "
              f"chol+orth2={d(41,40)} spmm_it3={d(8,41)} rest={d(11,8)} total={d(11,0)}  half products {np.bincount(st >> 8).nonzero()[0].tolist()}")
        print(f"      CSC list build: zero={d(55,1)} passA={d(50,55)} prefix={d(51,50)} class+perm={d(52,51)} scan+ptr={d(53,52)} passB={d(2,53)}")
    if kk <= 4:
        print(f"k={kk} (Gram path, products {np.bincount(st >> 8).nonzero()[0].tolist()}): stage={d(1,0)} group={d(2,1)} start={d(4,3)} Gzero={d(44,4)} pairs={d(45,44)} convert+Vinit={d(5,45)} iterate={d(11,6)} total={d(11,0)}  | group: zero={d(55,1)} passA={d(50,55)} scans={d(51,50)} passB={d(2,51)} | iteration 2: product={d(57,56)} sum+gram={d(58,57)} chol+stop={d(59,58)} orth={d(60,59)}")
PY
