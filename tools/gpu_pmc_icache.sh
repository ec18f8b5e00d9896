cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3n
rocprofv3 -L 2>/dev/null | grep -i -E "icache|ifetch|INST_CACHE|SQC_" | head -40 > gpurun_out/r3n/counters.txt
cat gpurun_out/r3n/counters.txt | cut -c1-160
CMD="python3 bench.py --steps 24 --warmup 4 --lanes 1 --no-cpu-baseline --spinup 0.2"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r3n/p1 -o p -- $CMD > gpurun_out/r3n/p1.log 2>&1 || tail -5 gpurun_out/r3n/p1.log
python3 tools/pmc_summarise.py $(find gpurun_out/r3n/p1 -name "*counter_collection.csv" | head -1) | grep -E "k_sparse_score" 
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/r3n/p2 -o p -- $CMD > gpurun_out/r3n/p2.log 2>&1 || tail -5 gpurun_out/r3n/p2.log
python3 tools/pmc_summarise.py $(find gpurun_out/r3n/p2 -name "*counter_collection.csv" | head -1) | grep -E "k_sparse_score"
find gpurun_out/r3n -name "*.csv" -size +3M -delete
