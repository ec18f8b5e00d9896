# Host-side AddressSanitizer run on the CPU box (no GPU calls; device code unchanged): builds libsplitp_hip_asan.so
# (make -C splitp_amd/csrc asan) and runs the ABI / host tests against it.  On the GPU box ROCm's ASan runtime
# intercepts the HSA allocator and wants xnack+ code objects, which this pool does not offer - so this stays a CPU check.
cd $(dirname $0)/..
[ -f splitp_amd/libsplitp_hip_asan.so ] || make -C splitp_amd/csrc asan > /dev/null 2>&1
export LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1
export SPLITP_LIB=$PWD/splitp_amd/libsplitp_hip_asan.so
python -m pytest tests/test_abi_and_host.py -q -x -p no:cacheprovider
