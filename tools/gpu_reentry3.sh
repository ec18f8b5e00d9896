# Final check of the re-entry session on HEAD (GPU box): GPU suite, smoke, the driver's command.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4f
mkdir -p $OUT
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log; grep -n "^E " $OUT/tests.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command_steps20.json 2> $OUT/bench_driver_command_steps20.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$OUT/bench_driver_command_steps20.json'))
print('driver command: value %.4g ms_per_step %.5f' % (d['value'], d['ms_per_step']), 'pipeline', d.get('north_star_pipeline',{}).get('ms_per_step'), 'roofline frac', d['roofline'].get('frac'), 'cpu', d['cpu_baseline']['value'])
"
