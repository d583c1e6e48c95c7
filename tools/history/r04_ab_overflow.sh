#!/bin/bash
OUT=gpurun_out/r04; mkdir -p $OUT
{
echo "## C2, overflow list on, one serialised launch per frame (pipeline_overflow 2), batch 8"; python3 tools/ab_kernels.py --option pipeline_overflow --values 2 --workload C2 --pipeline 1 --batch 8 --preset overflow_list=1 | tail -1
echo "## C2, overflow list on, two launches (pipeline_overflow 0)"; python3 tools/ab_kernels.py --option pipeline_overflow --values 0 --workload C2 --pipeline 1 --batch 8 --preset overflow_list=1 | tail -1
echo "## C3, serialised / two launches"; python3 tools/ab_kernels.py --option pipeline_overflow --values 2 0 --workload C3 --pipeline 1 --batch 8 --preset overflow_list=1 | tail -2
echo "## C2band, serialised / two launches"; python3 tools/ab_kernels.py --option pipeline_overflow --values 2 0 --workload C2band --pipeline 1 --batch 8 --preset overflow_list=1 | tail -2
} 2>&1 | grep -v amdgpu | tee $OUT/ab_overflow_serial.txt
