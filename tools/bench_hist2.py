"""Histogram forms side by side: direct 4^n bins vs radix sort + run-length encode (context option "hist_sort")."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splitp_amd as sp
from splitp_amd import synthetic as syn
for n, L in ((10, 100_000), (12, 4_000_000), (14, 8_000_000), (16, 1_000_000), (20, 1_000_000)):
    sites = syn.simulate_sites(n, L, 0.05, seed=2)
    sk = syn.site_keys(sites)
    for force in (("0", "1") if n <= 16 else ("1",)):
        sp.get_context().set_option("hist_sort", int(force))
        dev = sp.DeviceAlignment.from_site_keys(sk, n)
        dev.ctx.enable_timing(True)
        best = 1e9
        for _ in range(3):
            dev.ctx.reset_timing()
            t0 = time.perf_counter()
            dev = sp.DeviceAlignment.from_site_keys(sk, n)
            wall = time.perf_counter() - t0
            ph = dev.ctx.phase_times()
            best = min(best, sum(v[0] for k, v in ph.items() if k == "hist"))
        dev.ctx.enable_timing(False)
        print(f"n={n} L={L} D={dev.info()['D']} form={'sort+rle' if force == '1' else 'direct bins'}: device {best:.3f} ms, wall {wall*1e3:.2f} ms")
    sp.get_context().set_option("hist_sort", -1)
