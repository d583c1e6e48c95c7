#!/bin/bash
OUT=gpurun_out/exp8; mkdir -p $OUT
for WL in C2 C3; do
for G in 2048 4096 8192; do
python tools/ab_kernels.py --workload $WL --pipeline 0 --option commit_wide --values 0 1 --set integrate_grid=$G --frames 60 > $OUT/${WL}_g$G.log 2>&1
done
done
for f in $OUT/*.log; do echo "== $f"; grep "=" $f | grep -v amdgpu.ids | cut -c1-190; done
