set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o gpurun_out/mfma_f64_peak
gpurun_out/mfma_f64_peak | tee gpurun_out/mfma_f64_peak.txt
python tools/gpu_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/probe.txt
