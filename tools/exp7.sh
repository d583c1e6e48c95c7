#!/bin/bash
OUT=gpurun_out/exp7; mkdir -p $OUT
for WL in C2 C3; do
python tools/ab_kernels.py --workload $WL --pipeline 1 --option integrate_prefetch --values 0 1 --frames 60 > $OUT/${WL}_pipe_prefetch.log 2>&1
python tools/ab_kernels.py --workload $WL --pipeline 0 --option integrate_prefetch --values 0 1 --set integrate_grid=2048 --frames 60 > $OUT/${WL}_two_prefetch.log 2>&1
python tools/ab_kernels.py --workload $WL --pipeline 0 --option integrate_prefetch --values 0 1 --set integrate_grid=4096 --frames 60 > $OUT/${WL}_two_prefetch4k.log 2>&1
python tools/ab_kernels.py --workload $WL --pipeline 0 --option fused_plane --values 0 1 --frames 60 > $OUT/${WL}_two_plane.log 2>&1
done
for f in $OUT/*.log; do echo "== $f"; grep "=" $f | grep -v amdgpu.ids | cut -c1-190; done
