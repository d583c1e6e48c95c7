#!/bin/bash
# same-box A/B of the frame kernels: library of a commit (lib/alt/c_*.so) against the working tree
OUT=gpurun_out/r04; mkdir -p $OUT
{
for wl in C2 C3 C2band; do
  echo "## $wl pipelined batch 8"; tools/ab_commits.sh run --option lean_kernels --values 1 --workload $wl --pipeline 1 --batch 8
done
echo "## C3 two-launch"; tools/ab_commits.sh run --option lean_kernels --values 1 --workload C3 --pipeline 0
echo "## C2 two-launch"; tools/ab_commits.sh run --option lean_kernels --values 1 --workload C2 --pipeline 0
} 2>&1 | tee $OUT/ab_frame_${1:-x}.txt
