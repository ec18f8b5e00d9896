# PMC records of every workload bench.py reports a counter-based fraction for (GPU box), copied by the caller to
# profiles/rNN_pmc_binding_<workload>_<route>.json:  bash tools/gpu_pmc_all.sh TAG
cd $GRAFT_REPO_ROOT
TAG=${1:-r04}
run() { wl=$1; route=$2; shift 2; echo "== $wl $route"; PMC_STEPS=$1 PMC_WARMUP=$2 BENCH_ARGS="--workload $wl --route $route" timeout -k 10 500 bash tools/gpu_pmc_binding.sh ${TAG}_${wl}_${route} > gpurun_out/pmc_${TAG}_${wl}_${route}.log 2>&1; echo "rc=$?"; grep -c '"kernel"' gpurun_out/pmc_${TAG}_${wl}_${route}/binding.json; }
mkdir -p gpurun_out
run config2 auto 24 4
run config2 dense 12 2
run config5 auto 3 1
run config3 auto 10 2
run config4 auto 4 1
