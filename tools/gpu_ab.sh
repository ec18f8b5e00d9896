#!/bin/bash
# A/B of sparse.hip compile-time switches on the GPU box: bash tools/gpu_ab.sh "-DX=1" "-DX=0 -DY=2" ...
set -e
cd "$GRAFT_REPO_ROOT/splitp_amd/csrc"
for flags in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $flags -c sparse.hip -o /tmp/sparse_ab.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsplitp_hip.so api.o flatten.o gram.o gram_i8.o eigen.o /tmp/sparse_ab.o subflat.o hist.o divergence.o
  cd ../..
  echo "--- flags: $flags"
  python bench.py --no-cpu-baseline --steps 1500 --lanes 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lanes1 kernel_ms', round(d['roofline']['launch_ms'],5), 'value', round(d['value']))"
  python bench.py --no-cpu-baseline --steps 1500 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lanes3 value', round(d['value']))"
  cd splitp_amd/csrc
done
