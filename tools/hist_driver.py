"""One workload of the site-pattern histogram (parsers/fasta.py:48-63 on the device), repeated - the program profiled by
tools/gpu_hist_profile.sh:  python3 tools/hist_driver.py N_TAXA N_SITES REPS [sort|bins|auto] [three]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splitp_amd as sp
from splitp_amd import synthetic as syn
n, L, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
form = sys.argv[4] if len(sys.argv) > 4 else "auto"
ctx = sp.get_context()
ctx.set_option("hist_sort", {"sort": 1, "bins": 0, "auto": -1}[form])
if len(sys.argv) > 5 and sys.argv[5] == "three":
    ctx.set_option("sort_three_launch", 1)
sk = syn.site_keys(syn.simulate_sites(n, L, 0.05, seed=2))
dev = sp.DeviceAlignment.from_site_keys(sk, n)
ctx.enable_timing(True)
ctx.reset_timing()
for _ in range(reps):
    dev = sp.DeviceAlignment.from_site_keys(sk, n)
ms = ctx.phase_times()["hist"][0] / reps
print(f"hist n={n} L={L} D={dev.info()['D']} form={form}: {ms:.4f} ms of device time per alignment (HIP events), "
      f"{(8 if n > 16 else 4) * L / ms / 1e6:.1f} GB/s of site words read once")
