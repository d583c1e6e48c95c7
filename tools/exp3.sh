#!/bin/bash
OUT=gpurun_out/exp3; mkdir -p $OUT
for WL in C2 C3; do
for L in o0 o0s80 o2 o2s80; do
  VOXELHASH_LIB=$PWD/voxelhashing_demo_amd/lib/alt/lib_$L.so python tools/ab_kernels.py --workload $WL --pipeline 1 --option integrate_grid --values 128 256 512 --frames 60 > $OUT/${WL}_${L}.log 2>&1
done
done
for f in $OUT/*.log; do echo "== $f"; grep "=" $f | grep -v amdgpu.ids | cut -c1-160; done
