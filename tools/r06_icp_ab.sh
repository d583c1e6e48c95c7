#!/bin/bash
# same-box A/B of library variants of the one-launch Align by its own clock: mean period of rounds 1..18 (VH_ICP_STAMPS)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
ALT=voxelhashing_demo_amd/lib/alt
{ for i in 1 2 3; do
  for lib in $ALT/v_*.so voxelhashing_demo_amd/lib/libvoxelhash_hip.so; do
    [ -f "$lib" ] || continue
    echo -n "$(basename $lib): "
    VOXELHASH_LIB=$lib VH_ICP_STAMPS=1 timeout 300 python3 tools/icp_only.py 8 2>&1 | grep "next round" | awk '{n++; s+=$NF} END {printf "rounds %d  period %.3f us\n", n, s/n}'
  done
done; } > gpurun_out/r06_icp_ab.txt 2>&1
cat gpurun_out/r06_icp_ab.txt
