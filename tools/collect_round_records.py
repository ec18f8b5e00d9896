"""Copy what tools/gpu_round_records.sh TAG left under gpurun_out/ into profiles/ under the names bench.py and the docs
use (build container, after the GPU call):  python tools/collect_round_records.py r04"""
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
out = os.path.join(ROOT, "gpurun_out")
prof = os.path.join(ROOT, "profiles")
copied = []


def cp(src, dst):
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copyfile(src, os.path.join(prof, dst))
        copied.append(dst)
    else:
        print("missing or empty:", os.path.relpath(src, ROOT))


for d in sorted(glob.glob(os.path.join(out, f"pmc_{tag}_*"))):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)[len(f"pmc_{tag}_"):]          # <workload>_<route>
    cp(os.path.join(d, "binding.json"), f"{tag}_pmc_binding_{name}.json")
    cp(os.path.join(d, "kernel_stats.csv"), f"{tag}_kernel_stats_{name}.csv")
    cp(os.path.join(d, "passes.txt"), f"{tag}_pmc_passes_{name}.txt")
rec = os.path.join(out, f"records_{tag}")
for f in ("bench_driver_command_steps20", "bench_default", "bench_dropin_mode"):
    cp(os.path.join(rec, f + ".json"), f"{tag}_{f}.json")
for f in ("config2", "config2_dense", "config3", "config4", "config5", "dist1_alignments", "dist1_splits", "dist1_config4_splits"):
    cp(os.path.join(rec, "variants", f + ".json"), f"{tag}_bench_{f}.json")
for f in ("hist_forms", "hist_pmc_n16", "hist_pmc_n14", "big_table"):
    cp(os.path.join(rec, f + ".txt"), f"{tag}_{f}.txt")
print(len(copied), "files copied into profiles/")
