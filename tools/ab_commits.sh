#!/bin/bash
# Side-by-side timing of the library as it was at several commits, on ONE box in one call -- the only
# comparison that is not at the mercy of the +-5 % between boxes.  (Round 2 lost 10 % of the headline for a
# day to a change whose before / after had been measured on two different boxes.)
#   here (no GPU):   tools/ab_commits.sh build <commit>...      -> voxelhashing_demo_amd/lib/alt/c_<commit>.so
#   on the GPU box:  tools/ab_commits.sh run [ab_kernels.py arguments]
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
ALT=$ROOT/voxelhashing_demo_amd/lib/alt
if [ "${1:-}" = build ]; then
  shift; mkdir -p "$ALT"
  for c in "$@"; do
    T=$(mktemp -d)
    git -C "$ROOT" archive "$c" voxelhashing_demo_amd/csrc include | tar -x -C "$T" || exit 1
    make -s -C "$T/voxelhashing_demo_amd/csrc" OUTDIR="$T/out" "$T/out/libvoxelhash_hip.so" > "$T/build.log" 2>&1 || { tail "$T/build.log"; exit 1; }
    cp "$T/out/libvoxelhash_hip.so" "$ALT/c_$c.so"; rm -rf "$T"; echo "built c_$c.so"
  done
elif [ "${1:-}" = run ]; then
  shift
  for lib in "$ALT"/c_*.so "$ROOT/voxelhashing_demo_amd/lib/libvoxelhash_hip.so"; do
    [ -f "$lib" ] || continue
    echo -n "$(basename "$lib")  "
    VOXELHASH_LIB=$lib timeout 600 python3 "$ROOT/tools/ab_lib.py" "$@" 2>&1 | tail -1
  done
else
  echo "usage: $0 build <commit>... | run [ab_kernels.py arguments]"; exit 2
fi
