#!/bin/bash
# Wrapper for every gpurun call of this repo (build container):  tools/gpu.sh [--timeout S] -- '<command>'
# Exports the git commit (the .git directory does not travel to the GPU box) so that PMC records can be stamped with it.
commit=$(git -C "$(dirname "$0")/.." rev-parse --short=12 HEAD 2>/dev/null || echo unknown)
dirty=$(git -C "$(dirname "$0")/.." status --porcelain -- splitp_amd bench.py 2>/dev/null | grep -v '^??' | wc -l)
[ "$dirty" != "0" ] && commit="$commit+dirty"
args=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do args+=("$1"); shift; done
shift
exec /usr/local/graft/bin/gpurun "${args[@]}" -- "export SPLITP_GIT_COMMIT=$commit; $*"
