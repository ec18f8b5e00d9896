set -e
cd $GRAFT_REPO_ROOT
python tools/gpu_dbg.py 2>&1 | grep -E "bad count"
cd splitp_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSPK_HEAVY=60000 -c sparse.hip -o /tmp/sparse_nh.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsplitp_hip.so api.o flatten.o gram.o gram_i8.o eigen.o /tmp/sparse_nh.o subflat.o hist.o
cd ../..
echo "--- no heavy path"
python tools/gpu_dbg.py 2>&1 | grep -E "bad count"
