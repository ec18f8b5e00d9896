set -e
cd $GRAFT_REPO_ROOT
bash tools/variant_lib.sh sparse.hip /tmp/lib_stamps.so -DSPK_STAMPS
SPLITP_LIB=/tmp/lib_stamps.so python - <<'PY'
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
import torch
for k in (5, 4):
    sub = [s for s in splits if min(len(s[0]), len(s[1])) == k]
    taxa_arr, a_arr = batch.encode_splits(sub, dev, n)
    nrep = 8
    sc = torch.zeros(nrep * len(sub), dtype=torch.float64, device="cuda")
    st = torch.zeros(nrep * len(sub), dtype=torch.int32, device="cuda")
    for blk in (0, (nrep * len(sub)) // 2 + 3):
        lib.sp_debug_spk_stamp_block(blk)
        for rep in range(2):
            batch.score_encoded_multi_async([dev] * nrep, taxa_arr, a_arr, sc.data_ptr(), st.data_ptr())
            torch.cuda.synchronize()
        out = (C.c_longlong * 64)()
        lib.sp_debug_spk_stamps(out)
        o = np.array(out[:], dtype=np.int64)
        print(f"k={k} block {blk}: lists total {o[2]-o[1]}: zero {o[42]-o[1]} passA {o[43]-o[42]} prefix {o[46]-o[43]} classify {o[47]-o[46]} bucket scan {o[48]-o[47]} perm {o[49]-o[48]} offsets {o[52]-o[49]} passB {o[2]-o[52]}")
PY
