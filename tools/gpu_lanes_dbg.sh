#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
run() { python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-cpu-baseline "$@"; }
for i in 1 2 3 4 5 6; do
  run --lanes 2 > gpurun_out/bg_d_$i.json 2> gpurun_out/bg_d_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bg_d*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['launch_ms'],4))
PY
