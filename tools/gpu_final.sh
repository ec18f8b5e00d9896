# end-of-round verification (GPU box): bash tools/gpu_final.sh TAG
cd $GRAFT_REPO_ROOT
TAG=${1:-r03}
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
bash tools/gpu_pmc_all.sh $TAG 2>&1 | tail -10
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log; grep -n "^E " $OUT/tests.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
