cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for ln in 2 3 4 5 6 8; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --lanes $ln --no-cpu-baseline --no-pipeline-block > /tmp/b.json 2>/tmp/b.err || { tail -3 /tmp/b.err; continue; }
  python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print("lanes $ln steps 20: value %.4g ms_per_step %.5f host_us %.1f" % (d['value'], d['ms_per_step'], d.get('host_us_per_step', -1)))
PY
done
done
for ln in 3 4 6; do
  python bench.py --gpus 1 --steps 2000 --warmup 100 --lanes $ln --no-cpu-baseline --no-pipeline-block > /tmp/b.json 2>/tmp/b.err
  python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print("lanes $ln steps 2000: value %.4g ms_per_step %.5f host_us %.1f" % (d['value'], d['ms_per_step'], d.get('host_us_per_step', -1)))
PY
done
