set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -15
python tools/gpu_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/probe.txt
python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench2.json | cut -c1-1500
