# Extra counters of one bench command (GPU box): bash tools/gpu_pmc_extra.sh OUTDIR "COUNTERS ..." -- bench args
# e.g. bash tools/gpu_pmc_extra.sh gpurun_out/x "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" --workload config4
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$1; CTRS=$2; shift 2
mkdir -p $OUT
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/pass -o p -- python3 bench.py --steps 4 --warmup 1 --lanes 1 --no-cpu-baseline --no-pipeline-block --spinup 0.2 "$@" > $OUT/pass.log 2>&1 || { tail -5 $OUT/pass.log; exit 1; }
F=$(find $OUT/pass -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summarise.py "$F" | grep -E "subscore|sparse_score|sparse_slow|k_eig4"
find $OUT -name "*.csv" -size +3M -delete || true
find $OUT -name "*.db" -delete || true
