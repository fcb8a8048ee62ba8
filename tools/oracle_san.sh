#!/bin/bash
# The CPU oracle (test infrastructure) under AddressSanitizer + UndefinedBehaviorSanitizer: builds oracle/libwx_oracle_san.so and runs the
# CPU suite with it.  The sanitizer runtime must be the first library of the process, hence the preload; python's own
# allocations are not what is looked at, so leak detection is off.  usage: tools/oracle_san.sh [log]   (default profiles/r06_oracle_asan.log)
set -e
cd "$(dirname "$0")/.."
LOG=${1:-profiles/r06_oracle_asan.log}
make -C oracle libwx_oracle_san.so >/dev/null
ASAN=$(gcc -print-file-name=libasan.so)
UBSAN=$(gcc -print-file-name=libubsan.so)
{
  echo "# oracle/libwx_oracle_san.so = wx_oracle.c -O1 -g -fsanitize=address,undefined; pytest -m 'not gpu and not perf' with it ($(date -u +%Y-%m-%d))"
  echo "# gcc: $(gcc --version | head -1)"
  WX_ORACLE_SAN=1 LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    python -m pytest tests -q -m "not gpu and not perf" -p no:cacheprovider 2>&1 | tail -25
} | tee "$LOG"
