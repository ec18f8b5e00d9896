#!/bin/bash
# 1-rank torchrun (RCCL) form of bench.py: lanes sweep, with and without the high-priority RCCL stream
set -e
cd "$GRAFT_REPO_ROOT"
run() { python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 300 --warmup 30 --no-cpu-baseline "$@"; }
for l in 1 2 3 4; do
  run --lanes $l > gpurun_out/bd_hi_$l.json 2> gpurun_out/bd_hi_$l.err
  run --lanes $l --no-hipri > gpurun_out/bd_lo_$l.json 2> gpurun_out/bd_lo_$l.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bd_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['launch_ms'],4))
PY
