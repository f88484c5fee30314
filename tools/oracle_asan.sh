#!/bin/bash
# The CPU restatement of OpenCV's Farneback path under AddressSanitizer + UBSan (sanitizers run on
# the CPU build only).  Builds oracle/libfbref_asan.so and runs the oracle's own test file against it.
set -e
cd "$(dirname "$0")/.."
make -C oracle asan >/dev/null
ASAN_LIB=$(gcc -print-file-name=libasan.so)
LD_PRELOAD="$ASAN_LIB" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
    FBREF_LIBRARY="$PWD/oracle/libfbref_asan.so" python3 -m pytest tests/test_oracle_farneback.py -x -q -p no:cacheprovider "$@"
