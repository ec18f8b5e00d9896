# dense route: gram_i8.hip built with GI_WAVES_PER_SIMD = $@ each, bench --route dense (GPU box)
cd $GRAFT_REPO_ROOT/splitp_amd/csrc
cp ../libsplitp_hip.so /tmp/lib_keep.so
for w in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DGI_WAVES_PER_SIMD=$w -c gram_i8.hip -o /tmp/gram_i8_$w.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsplitp_hip.so api.o flatten.o gram.o /tmp/gram_i8_$w.o eigen.o sparse.o sparse_big.o subflat.o hist.o divergence.o
  (cd ../.. && python bench.py --steps 40 --warmup 4 --no-cpu-baseline --route dense > /tmp/bd.json 2>/tmp/bd.err && python - <<PY
import json
d=json.load(open('/tmp/bd.json'))
print("waves/simd $w: ms_per_step %.4f phases %s" % (d['ms_per_step'], d['roofline']['phase_ms_per_step']))
PY
  ) || tail -3 /tmp/bd.err
done
cp /tmp/lib_keep.so ../libsplitp_hip.so
