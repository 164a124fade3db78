#!/bin/bash
# The CPU tier of the device code (tests/host/*.cpp: the blob extraction, the voting item, the tail geometry and the
# libstdc++ / glibc restatement, compiled for the host) under AddressSanitizer + UndefinedBehaviorSanitizer:
#   tools/host_sanitize.sh [pytest args]        (GPU sanitizers are not available on this pool; this is the CPU build)
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
ASAN=$(g++ -print-file-name=libasan.so)   # (the test libraries are loaded by python: the ASan runtime must come first)
rc=0
for t in tests/test_k1b_host.py tests/test_geometry_host.py tests/test_ddmath_host.py tests/test_vote_host.py; do
  MPE_HOST_CXXFLAGS="-g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer" \
  LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 \
    python -m pytest $t -x -q -p no:cacheprovider -s "$@" 2>&1 | grep -E "runtime error|ERROR: AddressSanitizer|passed|failed|error" || true
  [ ${PIPESTATUS[0]} -ne 0 ] && rc=1
done
exit $rc
