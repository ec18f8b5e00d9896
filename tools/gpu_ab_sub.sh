# A/B of compile-time variants of the subflattening score kernel: SUB_VARIANTS="-DX=1|-DY=2" bash tools/gpu_ab_sub.sh
set -e
cd $GRAFT_REPO_ROOT/splitp_amd/csrc
IFS='|' read -ra VARS <<< "${SUB_VARIANTS:-}"
for v in "" "${VARS[@]}"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $v -c subflat.hip -o /tmp/subflat_ab.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsplitp_hip.so api.o flatten.o gram.o gram_i8.o eigen.o sparse.o /tmp/subflat_ab.o hist.o divergence.o
  echo "== variant '$v'"
  (cd ../.. && python tools/gpu_cfg3.py 2>&1 | grep "^n=" | cut -c1-200)
done
