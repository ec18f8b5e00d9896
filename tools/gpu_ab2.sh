# A/B of sparse.hip builds on the GPU box: bash tools/gpu_ab2.sh "<extra flags A>" "<extra flags B>" [bench args]
# (variant builds live in /tmp and are selected with SPLITP_LIB; the shipped library is never touched)
set -e
cd $GRAFT_REPO_ROOT
bash tools/variant_lib.sh sparse.hip /tmp/lib_A.so $1
bash tools/variant_lib.sh sparse.hip /tmp/lib_B.so $2
for rep in 1 2; do
for v in A B; do
  for lanes in 1 3; do
    SPLITP_LIB=/tmp/lib_$v.so python bench.py --steps 2000 --warmup 50 --no-cpu-baseline --no-pipeline-block --lanes $lanes ${3:-} > /tmp/b.json 2>/tmp/b.err || { tail -3 /tmp/b.err; }
    python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print("$v lanes $lanes: ms_per_step %.5f launch_ms %.5f" % (d['ms_per_step'], d['roofline']['launch_ms']))
PY
  done
done
done
