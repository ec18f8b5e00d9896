# A/B of sparse.hip builds on the GPU box: bash tools/gpu_ab2.sh "<extra flags A>" "<extra flags B>" [bench args]
set -e
cd $GRAFT_REPO_ROOT/splitp_amd/csrc
cp ../libsplitp_hip.so /tmp/lib_orig.so
for v in A B; do
  if [ $v = A ]; then FL="$1"; else FL="$2"; fi
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $FL -c sparse.hip -o /tmp/sparse_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_$v.so api.o flatten.o gram.o gram_i8.o eigen.o /tmp/sparse_$v.o sparse_big.o subflat.o hist.o divergence.o
done
cd ../..
for rep in 1 2; do
for v in A B; do
  cp /tmp/lib_$v.so splitp_amd/libsplitp_hip.so
  for lanes in 1 3; do
    python bench.py --steps 2000 --warmup 50 --no-cpu-baseline --lanes $lanes ${3:-} > /tmp/b.json 2>/tmp/b.err || { tail -3 /tmp/b.err; }
    python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print("$v lanes $lanes: ms_per_step %.5f launch_ms %.5f" % (d['ms_per_step'], d['roofline']['launch_ms']))
PY
  done
done
done
cp /tmp/lib_orig.so splitp_amd/libsplitp_hip.so
