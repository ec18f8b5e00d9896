cd $GRAFT_REPO_ROOT
for g in 1 2 4; do for lanes in 3 4; do
  vals=""
  for rep in 1 2 3 4 5; do
    v=$(python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline-block --group $g --lanes $lanes 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f' % d['ms_per_step'])")
    vals="$vals $v"
  done
  echo "group $g lanes $lanes: ms_per_step (20 steps)$vals"
done; done
