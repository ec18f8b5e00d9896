# Re-entry verification of round 4, second part (GPU box): the GPU suite on the final Python side, the drop-in records.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4e
mkdir -p $OUT
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $OUT/tests2.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests2.log; grep -n "^E " $OUT/tests2.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 150 python tools/gpu_dropin_profile.py 2>&1 | grep -v amdgpu.ids > $OUT/dropin_profile3.txt; head -5 $OUT/dropin_profile3.txt
python bench.py --mode dropin --steps 200 --warmup 20 --no-cpu-baseline --no-pipeline-block > $OUT/bench_dropin_mode.json 2> $OUT/bench_dropin_mode.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$OUT/bench_dropin_mode.json')); print('dropin', d['dropin']['value'], d['dropin']['seconds'], 'ms_per_step', d['ms_per_step'])"
