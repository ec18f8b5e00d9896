set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 120 python tools/gpu_probe.py 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/probe_sparse.txt
