"""Config 5 shape (12 taxa, 100 k sites): time of the sparse route per split size class (one alignment, synchronous call)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import splitp_amd as sp
from splitp_amd import synthetic as syn, batch, _lib
from splitp_amd import simulation as sim
import torch
n = 12
names = syn.taxa_names(n)
dev = sim.generate_device_alignment(syn.balanced_tree(n), sim.JukesCantor(), 100_000, seed=101, branch_length=0.05)
dev.taxa = tuple(names)
taxa_arr, a_arr = sp.encode_all_splits(n)
k = np.minimum(a_arr, n - a_arr)
print("patterns", len(dev))
for kk in (2, 3, 4, 5, 6):
    idx = np.nonzero(k == kk)[0]
    t, a = np.ascontiguousarray(taxa_arr[idx]), np.ascontiguousarray(a_arr[idx])
    batch.score_encoded(dev, t, a, _lib.SP_METHOD_FLATTENING)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        sc, st = batch.score_encoded(dev, t, a, _lib.SP_METHOD_FLATTENING)
    dt = (time.perf_counter() - t0) / 5
    print(f"k={kk}: {len(idx)} splits, {dt*1e3:.3f} ms per call = {dt/len(idx)*1e6:.2f} us per split, half products {np.bincount(st >> 8).nonzero()[0].tolist()}")

# saturated: 32 alignments x one size class per call (the shape of BASELINE config 5)
devs = []
for a in range(32):
    d = sim.generate_device_alignment(syn.balanced_tree(n), sim.JukesCantor(), 100_000, seed=200 + a, branch_length=0.05)
    d.taxa = tuple(names)
    devs.append(d)
tot = 0.0
for kk in (2, 3, 4, 5, 6):
    idx = np.nonzero(k == kk)[0]
    t, a = np.ascontiguousarray(taxa_arr[idx]), np.ascontiguousarray(a_arr[idx])
    sc = torch.zeros(32 * len(idx), dtype=torch.float64, device="cuda")
    st = torch.zeros(32 * len(idx), dtype=torch.int32, device="cuda")
    batch.score_encoded_multi_async(devs, t, a, sc.data_ptr(), st.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        batch.score_encoded_multi_async(devs, t, a, sc.data_ptr(), st.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    tot += dt
    print(f"saturated k={kk}: 32 x {len(idx)} items in {dt*1e3:.2f} ms = {dt/(32*len(idx))*1e6*256:.1f} us of one CU per item")
print(f"sum {tot*1e3:.1f} ms")
