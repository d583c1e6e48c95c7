#!/bin/bash
# Library variants built with different -D switches, timed side by side on ONE box.
#   here (no GPU):   tools/ab_variants.sh build name1 "-DX=1 -DY=2" name2 "..." ...   -> voxelhashing_demo_amd/lib/alt/v_<name>.so
#   on the GPU box:  tools/ab_variants.sh run <python script + arguments>            (each variant, then the in-tree library)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
ALT=$ROOT/voxelhashing_demo_amd/lib/alt
if [ "${1:-}" = build ]; then
  shift; mkdir -p "$ALT"; rm -f "$ALT"/v_*.so
  while [ $# -ge 2 ]; do
    name=$1; flags=$2; shift 2
    T=$(mktemp -d)
    make -s -C "$ROOT/voxelhashing_demo_amd/csrc" OUTDIR="$T" EXTRA="$flags" "$T/libvoxelhash_hip.so" > "$T/build.log" 2>&1 || { tail "$T/build.log"; exit 1; }
    cp "$T/libvoxelhash_hip.so" "$ALT/v_$name.so"; rm -rf "$T"; echo "built v_$name.so ($flags)"
  done
elif [ "${1:-}" = run ]; then
  shift
  for lib in "$ALT"/v_*.so "$ROOT/voxelhashing_demo_amd/lib/libvoxelhash_hip.so"; do
    [ -f "$lib" ] || continue
    echo "== $(basename "$lib")"
    VOXELHASH_LIB=$lib timeout 600 python3 "$@" 2>&1 | grep -v amdgpu.ids | tail -8
  done
else
  echo "usage: $0 build name flags ... | run script args"; exit 2
fi
