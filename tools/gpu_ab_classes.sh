cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in head tree; do
  if [ $v = head ]; then export SPLITP_LIB=$GRAFT_REPO_ROOT/tools/experiments/lib_head.so; else unset SPLITP_LIB; fi
  echo "== $v"; python tools/gpu_cfg2_classes.py 2>&1 | grep "^k="
done
done
