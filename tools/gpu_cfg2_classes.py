"""Config 2 shape (10 taxa, 100 k sites): saturated time of the sparse route per split size class (32 copies of the
alignment x one size class per call, as tools/gpu_cfg5_classes.py does for config 5) - shows which class a kernel change
moved, which the whole-launch benchmark cannot."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import splitp_amd as sp
from splitp_amd import synthetic as syn, batch, _lib
import torch
n, L = 10, 100_000
names = syn.taxa_names(n)
keys, counts = syn.pattern_table(syn.simulate_sites(n, L, 0.05, seed=1))
dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=L, taxa=names)
taxa_arr, a_arr = sp.encode_all_splits(n)
k = np.minimum(a_arr, n - a_arr)
devs = [dev] * 32
tot = 0.0
for kk in (2, 3, 4, 5):
    idx = np.nonzero(k == kk)[0]
    t, a = np.ascontiguousarray(taxa_arr[idx]), np.ascontiguousarray(a_arr[idx])
    sc = torch.zeros(32 * len(idx), dtype=torch.float64, device="cuda")
    st = torch.zeros(32 * len(idx), dtype=torch.int32, device="cuda")
    batch.score_encoded_multi_async(devs, t, a, sc.data_ptr(), st.data_ptr())
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(4):
            batch.score_encoded_multi_async(devs, t, a, sc.data_ptr(), st.data_ptr())
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 4)
    tot += best * len(idx) / len(idx)
    its = np.bincount(st.cpu().numpy() >> 8).nonzero()[0].tolist()
    print(f"k={kk}: 32 x {len(idx)} items in {best*1e3:.3f} ms = {best/(32*len(idx))*1e6*256:.2f} us of one CU per item, products {its}")
