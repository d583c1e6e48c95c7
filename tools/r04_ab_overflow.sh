#!/bin/bash
OUT=gpurun_out/r04; mkdir -p $OUT
{
echo "## C2, overflow list on, pipelined (serialised) batch 8"; tools/ab_commits.sh run --option lean_kernels --values 1 --workload C2 --pipeline 1 --batch 8 --preset overflow_list=1
echo "## C2, overflow list on, two launches"; tools/ab_commits.sh run --option lean_kernels --values 1 --workload C2 --pipeline 0 --preset overflow_list=1
echo "## C2band, overflow list on, two launches"; tools/ab_commits.sh run --option lean_kernels --values 1 --workload C2band --pipeline 0 --preset overflow_list=1
} 2>&1 | tee $OUT/ab_overflow_coop.txt
