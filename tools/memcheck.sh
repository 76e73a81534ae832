#!/bin/bash
# Memory-safety pass of the GPU suite (run on an MI355X box from the repo root):
#   1. TDK_DEBUG_CANARY=1: every device allocation between red zones, verified after every test
#   2. the host side of the library under AddressSanitizer (make -C tadataka_amd/csrc asan), CPU + GPU tests
#   3. the GPU suite with the HSA / HIP runtime's own fault reporting turned up (a device page fault aborts the
#      process with the faulting address; AMD_LOG_LEVEL=1 prints the runtime's errors)
# Writes gpurun_out/memcheck.txt (copied to profiles/r05_memcheck.txt).
OUT=gpurun_out/memcheck.txt
mkdir -p gpurun_out
{
echo "== 1. TDK_DEBUG_CANARY=1 python -m pytest tests -q -m gpu"
TDK_DEBUG_CANARY=1 python -m pytest tests -q -m gpu 2>&1 | tail -4
echo
echo "== 2. AddressSanitizer on the host side of libtadataka_hip (make asan), GPU suite + CPU suite"
make -s -C tadataka_amd/csrc asan -j8 > /dev/null 2>&1 || echo "make asan FAILED"
# libstdc++ beside it: the sanitizer resolves __cxa_throw when it starts, and python itself does not link libstdc++
RT="$(make -s -C tadataka_amd/csrc asan-runtime) $(readlink -f $(gcc -print-file-name=libstdc++.so))"
LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:log_path=gpurun_out/asan \
  TDK_LIBRARY=$PWD/tadataka_amd/lib/libtadataka_hip_asan.so python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -4
LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:log_path=gpurun_out/asan \
  TDK_LIBRARY=$PWD/tadataka_amd/lib/libtadataka_hip_asan.so python -m pytest tests -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -3
echo "ASAN reports written: $(ls gpurun_out/asan.* 2>/dev/null | wc -l)"
for f in gpurun_out/asan.*; do [ -f "$f" ] && { echo "--- $f"; head -30 "$f"; }; done
echo
echo "== 3. AMD_LOG_LEVEL=1 HSA_XNACK=0 python -m pytest tests -q -m gpu  (runtime errors and page faults reported)"
AMD_LOG_LEVEL=1 python -m pytest tests -q -m gpu 2> gpurun_out/amd_log.txt | tail -3
echo "lines of runtime log: $(wc -l < gpurun_out/amd_log.txt); lines mentioning a fault / violation: $(grep -ciE 'page fault|memory access fault|violation|HSA_STATUS_ERROR' gpurun_out/amd_log.txt)"
} > $OUT 2>&1
cat $OUT
