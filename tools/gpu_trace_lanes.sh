#!/bin/bash
# kernel timeline of bench.py with two lanes: are consecutive steps' kernels overlapped, and on which queues?
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trace_lanes -- python3 bench.py --steps 40 --warmup 10 --lanes 2 --no-cpu-baseline > gpurun_out/trace_lanes.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/trace_lanes/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print(rows[0].keys())
ks=[r for r in rows if 'sparse' in r['Kernel_Name']]
ks.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(ks[0]['Start_Timestamp'])
for r in ks[20:34]:
    print(r.get('Queue_Id'), r.get('Stream_Id'), (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
m=glob.glob('gpurun_out/trace_lanes/**/*memory_copy_trace.csv',recursive=True)
if m:
    rows=list(csv.DictReader(open(m[0])))
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    for r in rows[-8:]:
        print(r.get('Direction'), (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
PY
