# stamps (tools/gpu_stamps2.sh's table) for several builds of sparse.hip: bash tools/gpu_ab_stamps.sh "<flags A>" "<flags B>" ...
cd $GRAFT_REPO_ROOT
for fl in "$@"; do
  echo "=== flags: $fl"
  SPK_EXTRA="$fl" bash tools/gpu_stamps2.sh 2>&1 | grep -A1 "k=3 block 0\|k=2 block 0\|k=5 block 0\|k=4 block 0" | cut -c1-330
done
