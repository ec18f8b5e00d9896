# rocprofv3 evidence for bench.py's `roofline.binding` block (GPU box):  BENCH_ARGS="..." bash tools/gpu_pmc_binding.sh TAG
# Separate --pmc passes (never combined with runtime / sys traces), single lane so that launches do not overlap, then a
# --kernel-trace --stats pass of the same command.  Writes gpurun_out/pmc_TAG/{binding.json,kernel_stats.csv,passes.txt};
# binding.json carries the provenance (source hash, commit, command, workload shape) bench.py checks before using it.
set -e
cd $GRAFT_REPO_ROOT
TAG=${1:-r03}
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
CMD="python3 bench.py --steps ${PMC_STEPS:-24} --warmup ${PMC_WARMUP:-4} --lanes 1 --no-cpu-baseline --no-pipeline-block --spinup 0.2 ${BENCH_ARGS:-}"
export PMC_CMD="$CMD"
: > $OUT/passes.txt
i=0
for pass in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_ANY TCC_HIT_sum TCC_MISS_sum" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pass$i -o p -- $CMD > $OUT/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/pass$i.log; continue; }
  F=$(find $OUT/pass$i -name "*counter_collection.csv" | head -1)
  python3 tools/pmc_summarise.py "$F" >> $OUT/passes.txt
  echo "pass $i done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1 || tail -5 $OUT/trace.log
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv || true
python3 tools/pmc_summarise.py --json $OUT/passes.txt $OUT/kernel_stats.csv $OUT/trace.log > $OUT/binding.json
cat $OUT/binding.json
find $OUT -name "*.csv" -size +3M -delete || true
find $OUT -name "*.db" -delete || true
