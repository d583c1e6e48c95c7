#!/bin/bash
OUT=gpurun_out/exp2; mkdir -p $OUT
for WL in C2 C3; do
python tools/ab_kernels.py --workload $WL --pipeline 1 --option integrate_grid --values 256 512 1024 2048 --frames 60 > $OUT/${WL}_order1_igrid.log 2>&1
for L in order0 sgpr80 sgpr96; do
  VOXELHASH_LIB=$PWD/voxelhashing_demo_amd/lib/alt/lib_$L.so python tools/ab_kernels.py --workload $WL --pipeline 1 --option integrate_grid --values 512 2048 --frames 60 > $OUT/${WL}_${L}.log 2>&1
done
python tools/ab_kernels.py --workload $WL --pipeline 1 --option commit_blocks --values 32 128 --frames 60 > $OUT/${WL}_cblocks.log 2>&1
done
grep -h "=" $OUT/*.log | grep -v amdgpu.ids
for f in $OUT/*.log; do echo "== $f"; grep "=" $f | grep -v amdgpu.ids | cut -c1-160; done
