# build a sparse.hip variant with extra -D flags on the GPU box and time it.  usage: bash tools/gpu_variant.sh "<sed expr>" 
set -e
cd $GRAFT_REPO_ROOT
python tools/gpu_probe.py 3 2>&1 | grep "k=0"
cd splitp_amd/csrc
sed "$1" sparse.hip > /tmp/sparse_var.hip
sed -i 's/#include "common.h"/#include "'"$(pwd | sed 's/\//\\\//g')"'\/common.h"/' /tmp/sparse_var.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c /tmp/sparse_var.hip -o /tmp/sparse_var.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsplitp_hip.so api.o flatten.o gram.o gram_i8.o eigen.o /tmp/sparse_var.o subflat.o hist.o
cd ../..
echo "--- variant: $1"
python tools/gpu_probe.py 3 2>&1 | grep "k="
