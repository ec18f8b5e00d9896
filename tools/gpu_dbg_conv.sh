cd $GRAFT_REPO_ROOT/splitp_amd/csrc
cp ../libsplitp_hip.so /tmp/lib_keep.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DSPK_DEBUG_CONV -c sparse.hip -o /tmp/sparse_dbg.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsplitp_hip.so api.o flatten.o gram.o gram_i8.o eigen.o /tmp/sparse_dbg.o sparse_big.o subflat.o hist.o divergence.o
cd ../..
python tools/gpu_iters.py
cp /tmp/lib_keep.so splitp_amd/libsplitp_hip.so
