# Stage breakdown (s_memtime stamps) of the global-memory form on the 12-taxon 100 k-site table, per split shape.
set -e
cd $GRAFT_REPO_ROOT
cd splitp_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $SPK_EXTRA -DSPK_STAMPS -c sparse.hip -o /tmp/sparse_st.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsplitp_hip.so api.o flatten.o gram.o gram_i8.o eigen.o /tmp/sparse_st.o subflat.o hist.o divergence.o
cd ../..
python - <<'PY'
import sys, ctypes as C, numpy as np, time
sys.path.insert(0,'.')
import splitp_amd as sp
from splitp_amd import simulation as sim, synthetic as syn
n = 12
names = syn.taxa_names(n)
dev = sim.generate_device_alignment(syn.balanced_tree(n), sim.JukesCantor(), 100_000, seed=5, branch_length=0.05)
dev.taxa = tuple(names)
splits = list(sp.all_splits(names))
lib = dev.ctx._lib
lib.sp_debug_spk_stamps.argtypes = [C.POINTER(C.c_longlong)]
lib.sp_debug_spk_stamp_block(0)
for k in (6, 5, 4, 3):
    sub = [s for s in splits if min(len(s[0]), len(s[1])) == k][:256]
    sp.score_splits(dev, sub)
    t0 = time.perf_counter(); s, st = sp.score_splits(dev, sub, return_status=True); dt = time.perf_counter() - t0
    out = (C.c_longlong * 64)()
    lib.sp_debug_spk_stamps(out)
    o = np.array(out[:], dtype=np.int64)
    print(f"k={k} splits {len(sub)} wall {dt*1e3:.2f} ms its {sorted(set((np.asarray(st)>>8).tolist()))} | stage={o[1]-o[0]} CSC={o[2]-o[1]} CSR={o[3]-o[2]} start={o[4]-o[3]} init={o[7]-o[4]} "
          f"loop={o[11]-o[7]} total={o[11]-o[0]} (100 MHz ticks)")
PY
