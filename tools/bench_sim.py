"""Throughput of the device simulator + histogram: sites/s, wall clock around the whole call
(tree upload, k_simulate_sites, histogram, table compaction, one sync)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splitp_amd as sp
from splitp_amd import simulation as sim, synthetic as syn

jc = sim.JukesCantor()
for n, L in ((10, 100_000), (10, 10_000_000), (10, 100_000_000), (16, 1_000_000), (16, 20_000_000)):
    tree = syn.balanced_tree(n)
    sim.generate_device_alignment(tree, jc, 1000, seed=1, branch_length=0.05)          # warm
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        dev = sim.generate_device_alignment(tree, jc, L, seed=rep, branch_length=0.05)
        best = min(best, time.perf_counter() - t0)
    print(f"n={n} L={L}: {best*1e3:.2f} ms  -> {L/best/1e9:.2f} G sites/s, D={dev.info()['D']}")
