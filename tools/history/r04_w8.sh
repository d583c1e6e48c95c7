#!/bin/bash
OUT=gpurun_out/r04; mkdir -p $OUT
for wl in C2 C3 C2band; do
  echo "## $wl pipelined batch 8"
  tools/ab_variants.sh run tools/ab_kernels.py --option lean_kernels --values 1 --workload $wl --pipeline 1 --batch 8 ${1:-}
done 2>&1 | tee $OUT/ab_pipe_waves.txt
