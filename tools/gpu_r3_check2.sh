cd $GRAFT_REPO_ROOT
TAG=${1:-r3x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 300 bash tools/gpu_stamps_cfg5.sh > $OUT/stamps_cfg5.txt 2>&1; echo "stamps5 rc=$?"; grep -A1 "^k=" $OUT/stamps_cfg5.txt | cut -c1-420
timeout -k 10 300 bash tools/gpu_stamps2.sh > $OUT/stamps_cfg2.txt 2>&1; echo "stamps2 rc=$?"; grep -A1 "block 0" $OUT/stamps_cfg2.txt | cut -c1-330
bash tools/gpu_r3_check.sh $TAG $2
