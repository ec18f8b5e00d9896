"""Direct solver (finish.hip) on generic matrices of growing smaller side: wall time per matrix and the error against LAPACK."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import splitp_amd as sp
from oracle import splitp_oracle as O
ctx = sp.get_context()
ctx.set_option("direct_all", 1)
rng = np.random.default_rng(7)
for m, k in ((64, 200), (256, 400), (512, 700), (1024, 1500), (2048, 2500), (4096, 4500)):
    M = np.where(rng.random((m, k)) < 0.05, rng.integers(1, 200, (m, k)), 0).astype(np.float64)
    want = O.dense_split_score(M)
    sp.split_score(M)
    t0 = time.perf_counter()
    got = sp.split_score(M)
    dt = time.perf_counter() - t0
    print(f"m = {m:5d} x {k}: {dt*1e3:9.2f} ms (upload + fp64 Gram + {2*m} launches + Sturm), |score - LAPACK| = {abs(got - want):.2e}", flush=True)
