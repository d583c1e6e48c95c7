#!/bin/bash
OUT=gpurun_out/exp5; mkdir -p $OUT
for WL in C2 C3 C2band; do
python tools/ab_kernels.py --workload $WL --pipeline 1 --option pipe_wave_integrate --values 0 1 --set pipe_integrate_grid=256 --frames 60 > $OUT/${WL}_g256.log 2>&1
python tools/ab_kernels.py --workload $WL --pipeline 1 --option pipe_wave_integrate --values 0 1 --set pipe_integrate_grid=512 --frames 60 > $OUT/${WL}_g512.log 2>&1
python tools/ab_kernels.py --workload $WL --pipeline 1 --option pipe_wave_integrate --values 0 1 --set pipe_integrate_grid=1024 --frames 60 > $OUT/${WL}_g1024.log 2>&1
done
for f in $OUT/*.log; do echo "== $f"; grep "=" $f | grep -v amdgpu.ids | cut -c1-190; done
