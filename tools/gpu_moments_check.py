"""Moment matrix M = sum_p w(p) s(p) s(p)^T of the subflattening path: the matrix-core kernel (default up to 21 taxa) against
the round-1 vector kernels (option moments_valu = 1) and the oracle, count and float-weight tables, and its time (GPU box):
    python tools/gpu_moments_check.py"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import splitp_amd as sp
from splitp_amd import _lib, synthetic as syn
from oracle import splitp_oracle as O

ctx = sp.get_context()
lib = ctx._lib


def moments(keys, counts, n, length, exact, valu):
    ctx.set_option("moments_valu", valu)
    names = syn.taxa_names(n)
    dev = (sp.DeviceAlignment.from_arrays(keys, None, n, counts=counts, n_sites=length, taxa=names) if exact else
           sp.DeviceAlignment.from_arrays(keys, counts / float(length), n, taxa=names, exact=False))
    m = 3 * n + 1
    out_i = np.zeros(m * m, dtype=np.int64)
    out_f = np.zeros(m * m, dtype=np.float64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _lib.check(lib.sp_moment_matrix(dev.handle, _lib._ptr(out_i, C.c_int64) if exact else None,
                                    None if exact else _lib._ptr(out_f, C.c_double)))
    dt = time.perf_counter() - t0
    ctx.set_option("moments_valu", 0)
    return (out_i if exact else out_f).reshape(m, m), dt


for n, length, seed in ((4, 300, 1), (8, 20_000, 2), (12, 200_000, 3), (16, 1_000_000, 4), (20, 1_000_000, 5), (21, 50_000, 6)):
    sites = syn.simulate_sites(n + (n & 1), length, 0.05, seed=seed)[:, :n]
    keys, counts = syn.pattern_table(sites)
    want = O.moment_matrix(keys, counts, n)
    for exact in (True, False):
        new, t_new = moments(keys, counts, n, length, exact, 0)
        new2, t_new2 = moments(keys, counts, n, length, exact, 0)
        old, t_old = moments(keys, counts, n, length, exact, 1)
        if exact:
            ok = np.array_equal(new, want) and np.array_equal(old, want) and np.array_equal(new, new2)
            err = int(np.abs(new - want).max())
        else:
            ref = want / float(length)
            err = float(np.abs(new - ref).max())
            ok = err <= 1e-13 and float(np.abs(old - ref).max()) <= 1e-13 and np.array_equal(new, new2)
        print(f"n {n:2d} D {len(keys):7d} {'counts' if exact else 'float '}: matrix cores {t_new2 * 1e3:.3f} ms (first call {t_new * 1e3:.3f}), vector kernels {t_old * 1e3:.3f} ms (host wall incl. upload), max error {err}  {'ok' if ok else 'WRONG'}")
        if not ok:
            sys.exit(1)
print("OK")
