"""Histogram pass (alignment columns -> pattern table, csrc/hist.hip) throughput on the GPU box.
Reports time per phase and achieved GB/s against the algorithmic bytes (site keys read once: 8 B/site;
ASCII path: n bytes/site read + 8 B/site written + read again)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splitp_amd as sp
from splitp_amd import synthetic as syn

for n, L in ((10, 100_000), (10, 4_000_000), (12, 4_000_000), (14, 8_000_000)):
    sites = syn.simulate_sites(n, L, 0.05, seed=2)
    keys = syn.site_keys(sites)
    ctx = sp.get_context()
    dev = sp.DeviceAlignment.from_site_keys(keys, n)          # warm-up (allocations)
    ctx.enable_timing(True); ctx.reset_timing()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        dev = sp.DeviceAlignment.from_site_keys(keys, n)
    wall = (time.perf_counter() - t0) / reps
    ph = ctx.phase_times(); ctx.enable_timing(False)
    hist_ms = ph["hist"][0] / reps
    uk, cnt = syn.pattern_table(sites)
    gk, gw, gc = dev.fetch()
    assert np.array_equal(gk, uk) and np.array_equal(gc, cnt)
    bins_bytes = 4.0 * 4 ** n
    print(f"n={n} L={L} D={len(uk)}: device histogram {hist_ms:.3f} ms (memset {bins_bytes/1e6:.0f} MB bins + hist + count + scan + write), "
          f"wall incl. host narrowing + H2D of keys {wall*1e3:.2f} ms; algorithmic bytes = {4 if n <= 15 else 8}L + 3*4*4^n "
          f"(bins zeroed, counted, compacted) = {((4 if n <= 15 else 8)*L + 3*bins_bytes)/1e6:.1f} MB -> "
          f"{((4 if n <= 15 else 8)*L + 3*bins_bytes)/hist_ms/1e6:.1f} GB/s")
