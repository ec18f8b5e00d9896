# round-3 baseline evidence before any kernel change (GPU box): bash tools/gpu_round3_baseline.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3a/tests.log
PMC_STEPS=3 PMC_WARMUP=1 BENCH_ARGS="--workload config5" timeout -k 10 500 bash tools/gpu_pmc_binding.sh r3a_config5 > gpurun_out/r3a/pmc5.log 2>&1; echo "pmc5 rc=$?"
PMC_STEPS=10 PMC_WARMUP=2 BENCH_ARGS="--workload config3" timeout -k 10 400 bash tools/gpu_pmc_binding.sh r3a_config3 > gpurun_out/r3a/pmc3.log 2>&1; echo "pmc3 rc=$?"
PMC_STEPS=4 PMC_WARMUP=1 BENCH_ARGS="--workload config4" timeout -k 10 400 bash tools/gpu_pmc_binding.sh r3a_config4 > gpurun_out/r3a/pmc4.log 2>&1; echo "pmc4 rc=$?"
timeout -k 10 200 python tools/gpu_cfg5_classes.py > gpurun_out/r3a/cfg5_classes.txt 2>&1; echo "classes rc=$?"; cat gpurun_out/r3a/cfg5_classes.txt | grep -v amdgpu.ids
timeout -k 10 300 bash tools/gpu_stamps_cfg5.sh > gpurun_out/r3a/cfg5_stamps.txt 2>&1; echo "stamps rc=$?"; grep "^k=" gpurun_out/r3a/cfg5_stamps.txt
