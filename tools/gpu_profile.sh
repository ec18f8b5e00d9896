# rocprofv3 kernel-trace + stats of the benchmark command (GPU box).  Usage: bash tools/gpu_profile.sh TAG [bench args]
set -e
cd $GRAFT_REPO_ROOT
TAG=${1:-r01}; shift || true
mkdir -p gpurun_out/prof_$TAG
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/prof_$TAG/bench_stdout.txt 2>&1 || (tail -20 gpurun_out/prof_$TAG/bench_stdout.txt; exit 1)
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -3
F=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp "$F" gpurun_out/${TAG}_kernel_stats.csv
head -20 "$F"
tail -2 gpurun_out/prof_$TAG/bench_stdout.txt
# keep the merge small: drop the raw per-dispatch trace
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete || true
