set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o gpurun_out/mfma_f64_peak 2>&1 | tail -3
gpurun_out/mfma_f64_peak | tee gpurun_out/mfma_f64_peak.txt
python -m pytest tests -x -q -m gpu 2>&1 | tail -5
python __graft_entry__.py smoke 2>&1 | tail -2
python bench.py --steps 100 --warmup 10 2>&1 | tee gpurun_out/bench1.json | tail -3
