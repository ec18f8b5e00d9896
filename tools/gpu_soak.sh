# Soak of the randomised sweeps + determinism tools on the current build (GPU box): bash tools/gpu_soak.sh OUTFILE [scale]
# Fresh seeds per round (pass a different BASE); default chain, LDS capped at 30 / 12 KB (hand-back chain on small tables),
# forced big-table form with both compactions, subflattening, generic matrices, mutual information, determinism.
cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/soak.txt}
B=${2:-31000}
mkdir -p $(dirname $OUT)
: > $OUT
run() { echo "== $*" >> $OUT; timeout -k 10 280 "$@" 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-6} >> $OUT; echo "   rc=$?" >> $OUT; echo "done: $*"; }
run python tools/gpu_fuzz_long.py $((B+1)) 2500 12
run python tools/gpu_fuzz_long.py $((B+2)) 1500 14
SPLITP_DEBUG_LDS_CAP=30000 run python tools/gpu_fuzz_long.py $((B+3)) 1500 12
SPLITP_DEBUG_LDS_CAP=12000 run python tools/gpu_fuzz_long.py $((B+4)) 1500 12
SPLITP_FORCE_BIG=1 run python tools/gpu_fuzz_long.py $((B+5)) 1200 12
SPLITP_FORCE_BIG=1 SPLITP_BIG_BY_KEYS=1 run python tools/gpu_fuzz_long.py $((B+6)) 800 12
run python tools/gpu_fuzz_sub.py $((B+7)) 1500
run python tools/gpu_fuzz_matrix.py $((B+8)) 1500
run python tools/gpu_fuzz_mi.py $((B+9)) 600
run python tools/gpu_determinism.py $((B+10)) 300 10
run python tools/gpu_determinism_xproc.py $((B+11)) 400 12 /tmp/xproc_a.npy
run python tools/gpu_determinism_xproc.py $((B+11)) 400 12 /tmp/xproc_b.npy
python - >> $OUT <<'PY'
import numpy as np
a, b = np.load("/tmp/xproc_a.npy"), np.load("/tmp/xproc_b.npy")
print("cross-process determinism: %d scores, %d differ bitwise" % (a.size, int((a.view(np.uint64) != b.view(np.uint64)).sum())))
PY
cat $OUT | grep -E "^==|^seed|BAD|EXC|differ|rc=|cross-process" | head -80
