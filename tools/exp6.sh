#!/bin/bash
OUT=gpurun_out/exp6; mkdir -p $OUT
for WL in C2 C3; do
python tools/ab_kernels.py --workload $WL --pipeline 1 --option pipe_integrate_grid --values 256 512 1024 2048 --frames 60 > $OUT/${WL}_pipe.log 2>&1
python tools/ab_kernels.py --workload $WL --pipeline 0 --option integrate_grid --values 512 1024 2048 4096 --frames 60 > $OUT/${WL}_two.log 2>&1
done
for f in $OUT/*.log; do echo "== $f"; grep "=" $f | grep -v amdgpu.ids | cut -c1-190; done
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_parity.py -q -x 2>&1 | tail -3
