# Build a VARIANT of the library without touching the shipped one (GPU box or build container):
#   bash tools/variant_lib.sh <source.hip> <out.so> [extra hipcc flags ...]
# compiles splitp_amd/csrc/<source.hip> with the extra flags into /tmp and links it with the tree's other objects into
# <out.so>; select it with SPLITP_LIB=<out.so> (splitp_amd/_lib.py).  ADVICE r2: the A/B tools used to overwrite
# splitp_amd/libsplitp_hip.so and restore it on their last line - a failed step left a diagnostic build in place.
set -e
src=$1; out=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
csrc=$root/splitp_amd/csrc
obj=/tmp/variant_$(basename $src .hip)_$$.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function "$@" -c $csrc/$src -o $obj
others=""
for f in api flatten gram gram_i8 eigen sparse sparse_big subflat subflat_pair hist divergence finish eig4 node; do
  if [ "$f.hip" = "$src" ]; then others="$others $obj"; else others="$others $csrc/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $others
rm -f $obj
