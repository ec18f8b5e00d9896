#!/bin/bash
# lanes / alignments sweep of bench.py plus the 1-rank torchrun (RCCL) form of the same command
set -e
cd "$GRAFT_REPO_ROOT"
for l in 1 2 3; do for a in 1 4; do
  python bench.py --steps 300 --warmup 30 --lanes $l --alignments $a --no-cpu-baseline > gpurun_out/bl_${l}_${a}.json
done; done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/bl_dist1.json 2> gpurun_out/bl_dist1.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bl_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['launch_ms'],4))
PY
