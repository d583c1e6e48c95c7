#!/bin/bash
# The oracle under AddressSanitizer + UndefinedBehaviorSanitizer: the CPU side of the repository is the only place where
# sanitizers can run (GPU ASan / XNACK are not available on the pool).  Builds oracle/_build/libvh_oracle.so with
# -fsanitize=address,undefined (no recovery from UB), runs the whole `-m "not gpu"` suite against it with libasan
# preloaded, then rebuilds the normal library.  Exit code = pytest's.
#   tools/sanitize_oracle.sh [extra pytest arguments]
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT" || exit 2
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g"
make -s -C oracle clean >/dev/null 2>&1
make -s -C oracle all EXTRA="$SAN" || { echo "sanitizer build failed"; exit 2; }
LIBASAN=$(gcc -print-file-name=libasan.so)
# detect_leaks=0: the interpreter itself leaks at exit; everything the oracle allocates is freed by vho_destroy, which the
# suite calls (OracleTable.close) -- and a heap overflow or use-after-free still aborts the run.
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  LD_PRELOAD="$LIBASAN" python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider "$@"
rc=$?
make -s -C oracle clean >/dev/null 2>&1
make -s -C oracle all || exit 2
echo "sanitize_oracle: pytest exit $rc (oracle rebuilt without sanitizers)"
exit $rc
