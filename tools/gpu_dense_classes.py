"""Dense route (north-star pipeline) at config 2 shape: per split size class, the phase times of one sp_score_splits call
and the histogram of products the certified 4-wide eigen kernel (eig4.hip) needed; then the whole list."""
import sys
import numpy as np
sys.path.insert(0, '.')
import splitp_amd as sp
from splitp_amd import synthetic as syn, batch, _lib
n, L = 10, 100_000
names = syn.taxa_names(n)
keys, counts = syn.pattern_table(syn.simulate_sites(n, L, 0.05, seed=1))
dev = sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=L, taxa=names)
taxa_arr, a_arr = sp.encode_all_splits(n)
k = np.minimum(a_arr, n - a_arr)
ctx = dev.ctx
for kk in (2, 3, 4, 5, 0):
    idx = np.nonzero(k == kk)[0] if kk else np.arange(len(k))
    t, a = np.ascontiguousarray(taxa_arr[idx]), np.ascontiguousarray(a_arr[idx])
    for rep in range(3):
        batch.score_encoded(dev, t, a, _lib.SP_METHOD_FLATTENING_DENSE)
    ctx.enable_timing(True)
    ctx.reset_timing()
    reps = 10
    for rep in range(reps):
        sc, st = batch.score_encoded(dev, t, a, _lib.SP_METHOD_FLATTENING_DENSE)
    pt = ctx.phase_times()
    ctx.enable_timing(False)
    its = np.bincount(st >> 8)
    ph = {name: round(v[0] / reps, 4) for name, v in pt.items() if v[0] > 0}
    print(f"k={kk or 'all'}: {len(idx)} splits, phases ms {ph}, products histogram {dict((i, int(c)) for i, c in enumerate(its) if c)}, flagged {int(np.count_nonzero(st & 3))}")
