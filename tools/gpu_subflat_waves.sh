# Workgroup shape of the batched subflattening score (option "subscore_waves"), configs 3 and 4:  bash tools/gpu_subflat_waves.sh
cd $GRAFT_REPO_ROOT
set -e
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "subflat or config3 or config4 or kernels_agree" 2>&1 | tail -3
for rep in ${REPS:-1 2}; do
for wv in 4 8 12 16 0; do
  export SPLITP_SUBSCORE_WAVES=$wv
  for wl in config3 config4; do
    steps=10; [ $wl = config3 ] && steps=100
    timeout -k 10 300 python bench.py --workload $wl --steps $steps --warmup 3 --no-cpu-baseline --no-pipeline-block > /tmp/b.json 2>/tmp/b.err || { tail -3 /tmp/b.err; exit 1; }
    python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print("waves $wv $wl: ms_per_step %.4f launch_ms %.4f" % (d['ms_per_step'], d['roofline']['launch_ms']))
PY
  done
done
done
