#!/bin/bash
OUT=gpurun_out/exp4; mkdir -p $OUT
W=$PWD/voxelhashing_demo_amd/lib/alt/lib_wg.so
for WL in C2 C3 C2band; do
python tools/ab_kernels.py --workload $WL --pipeline 1 --option pipe_integrate_grid --values 128 256 512 1024 --frames 60 > $OUT/${WL}_wave_pipe.log 2>&1
VOXELHASH_LIB=$W python tools/ab_kernels.py --workload $WL --pipeline 1 --option pipe_integrate_grid --values 512 2048 --frames 60 > $OUT/${WL}_wg_pipe.log 2>&1
python tools/ab_kernels.py --workload $WL --pipeline 0 --option integrate_grid --values 256 512 1024 2048 --frames 60 > $OUT/${WL}_wave_two.log 2>&1
VOXELHASH_LIB=$W python tools/ab_kernels.py --workload $WL --pipeline 0 --option integrate_grid --values 1024 2048 4096 --frames 60 > $OUT/${WL}_wg_two.log 2>&1
done
python tools/ab_kernels.py --workload C3 --pipeline 1 --option pipe_wide --values 0 1 --frames 60 > $OUT/C3_wide.log 2>&1
python tools/ab_kernels.py --workload C2 --pipeline 1 --option pipe_wide --values 0 1 --frames 60 > $OUT/C2_wide.log 2>&1
for f in $OUT/*.log; do echo "== $f"; grep "=" $f | grep -v amdgpu.ids | cut -c1-190; done
