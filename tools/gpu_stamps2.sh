set -e
cd $GRAFT_REPO_ROOT
bash tools/variant_lib.sh sparse.hip /tmp/lib_stamps.so $SPK_EXTRA -DSPK_STAMPS
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
for k in (5, 4, 3, 2):
    sub = [s for s in splits if min(len(s[0]), len(s[1])) == k]
    taxa_arr, a_arr = batch.encode_splits(sub, dev, n)
    nrep = 8
    sc = torch.zeros(nrep * len(sub), dtype=torch.float64, device="cuda")
    st = torch.zeros(nrep * len(sub), dtype=torch.int32, device="cuda")
    for blk in (0, (nrep * len(sub)) // 2 + 3, nrep * len(sub) - 1):
        lib.sp_debug_spk_stamp_block(blk)
        for rep in range(2):
            batch.score_encoded_multi_async([dev] * nrep, taxa_arr, a_arr, sc.data_ptr(), st.data_ptr())
            torch.cuda.synchronize()
        out = (C.c_longlong * 64)()
        lib.sp_debug_spk_stamps(out)
        o = np.array(out[:], dtype=np.int64)
        # (every number printed is the difference of two stamps the path in question really sets: a general-path split
        # sets none of the Gram path's stamps and vice versa - round 2's table printed those too, as garbage)
        if k >= 4:
            print(f"k={k} block {blk} (general path): stage={o[1]-o[0]} both lists={o[2]-o[1]} start rows={o[4]-o[3]} W init + first half product + orth={o[7]-o[4]} "
                  f"spmm_it2={o[9]-o[7]} gram2={o[40]-o[9]} orth2={o[41]-o[40]} spmm_it3={o[8]-o[41]} rest={o[11]-o[8]} total={o[11]-o[0]}")
            print(f"     spmm_it3 by size class: wave={o[30]-o[41]} row={o[32]-o[30]} quad={o[20]-o[32]} lane={o[8]-o[20]}")
        else:
            print(f"k={k} block {blk} (Gram path): stage={o[1]-o[0]} list={o[2]-o[1]} start rows={o[4]-o[3]} G zero={o[44]-o[4]} pairs={o[45]-o[44]} "
                  f"convert + V init={o[5]-o[45]} iterate={o[11]-o[6]} total={o[11]-o[0]} | iteration 2: product={o[57]-o[56]} sum+gram={o[58]-o[57]} chol+stop={o[59]-o[58]} orth={o[60]-o[59]}")
PY
