"""cProfile of bench.py --mode dropin (GPU box): the same README loop as tools/gpu_dropin_profile.py, but inside the
bench process (lanes, streams and pinned buffers of the main region alive) - where the factor 2 between the two goes."""
import cProfile, io, pstats, sys
sys.path.insert(0, '.')
sys.argv = ["bench.py", "--mode", "dropin", "--steps", "200", "--warmup", "20", "--no-cpu-baseline", "--no-pipeline-block"]
import bench
pr = cProfile.Profile()
orig = bench.dropin_loop
def wrapped(*a, **k):
    pr.enable()
    try:
        return orig(*a, **k)
    finally:
        pr.disable()
bench.dropin_loop = wrapped
bench.main()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(24); print(s.getvalue()[:5000])
