# dense route A/B on the GPU box: int8 Gram on 64 x 64 tiles (SPLITP_GRAM_TILE64=1) against the default 128 x 128 tiles
cd $GRAFT_REPO_ROOT
for v in ${AB_VALUES:-1 0}; do
  env ${AB_VAR:-SPLITP_GRAM_TILE64}=$v python bench.py --steps 40 --warmup 4 --no-cpu-baseline --route dense > /tmp/bd.json 2>/tmp/bd.err && python - <<PY || tail -3 /tmp/bd.err
import json
d=json.load(open('/tmp/bd.json'))
print("${AB_VAR:-SPLITP_GRAM_TILE64}=$v: ms_per_step %.4f phases %s" % (d['ms_per_step'], d['roofline']['phase_ms_per_step']))
PY
done
