# Everything profiles/ records for a round, on the FINAL build (GPU box): bash tools/gpu_round_records.sh r04
# 1. PMC + kernel statistics of every workload (hash-stamped), 2. the driver's command with CPU legs, 3. the other bench
# workloads, 4. histogram forms + per-kernel HBM bytes of the histogram pass, 5. the drop-in loop.
cd $GRAFT_REPO_ROOT
TAG=${1:-r04}
OUT=gpurun_out/records_$TAG
mkdir -p $OUT
bash tools/gpu_pmc_all.sh $TAG 2>&1 | tail -12
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command_steps20.json 2> $OUT/bench_driver_command_steps20.err
python bench.py --steps 2000 --warmup 100 --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --mode dropin --steps 200 --warmup 20 --no-cpu-baseline --no-pipeline-block > $OUT/bench_dropin_mode.json 2> $OUT/bench_dropin_mode.err
bash tools/gpu_bench_variants.sh $OUT/variants 2>&1 | grep -v amdgpu.ids | grep -E "^==|value"
timeout -k 10 120 python tools/bench_hist2.py 2>&1 | grep -v amdgpu > $OUT/hist_forms.txt; cat $OUT/hist_forms.txt
bash tools/gpu_hist_profile.sh $OUT/hist_n16 16 1000000 sort 2>&1 | grep -v amdgpu.ids > $OUT/hist_pmc_n16.txt; tail -4 $OUT/hist_pmc_n16.txt
bash tools/gpu_hist_profile.sh $OUT/hist_n14 14 8000000 sort 2>&1 | grep -v amdgpu.ids > $OUT/hist_pmc_n14.txt; tail -4 $OUT/hist_pmc_n14.txt
timeout -k 10 100 python tools/gpu_big_time.py 2>&1 | grep splits > $OUT/big_table.txt; cat $OUT/big_table.txt
