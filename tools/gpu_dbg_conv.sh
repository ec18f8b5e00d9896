# per-half-product convergence dump of the sparse kernel (-DSPK_DEBUG_CONV build in /tmp, selected with SPLITP_LIB)
set -e
cd $GRAFT_REPO_ROOT
bash tools/variant_lib.sh sparse.hip /tmp/lib_dbg.so -DSPK_DEBUG_CONV
SPLITP_LIB=/tmp/lib_dbg.so python tools/gpu_iters.py
