#!/bin/bash
# kernel + copy timeline of the 1-rank RCCL form of bench.py with two lanes
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trace_dist -- python3 bench.py --steps 40 --warmup 10 --lanes ${1:-2} --no-cpu-baseline > gpurun_out/trace_dist.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/trace_dist/**/*kernel_trace.csv',recursive=True)[0]
ev=[]
for r in csv.DictReader(open(f)):
    ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'K q%s s%s %s'%(r['Queue_Id'],r['Stream_Id'],r['Kernel_Name'][:40])))
m=glob.glob('gpurun_out/trace_dist/**/*memory_copy_trace.csv',recursive=True)
if m:
    for r in csv.DictReader(open(m[0])):
        ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'C '+r['Direction']))
ev.sort()
sp=[e for e in ev if 'sparse' in e[2]]
t0=sp[25][0]
for e in ev:
    if e[0]>=t0 and e[0]<t0+2.2e6:
        print('%9.1f %9.1f %7.1f %s'%((e[0]-t0)/1e3,(e[1]-t0)/1e3,(e[1]-e[0])/1e3,e[2]))
PY
tail -2 gpurun_out/trace_dist.log | cut -c1-300
