#!/bin/bash
# experiment: C3 walk variants and entries-per-lane builds
OUT=gpurun_out/exp1; mkdir -p $OUT
python tools/ab_kernels.py --workload C3 --option flatten_variant --values 3 5 --frames 60 > $OUT/c3_variant.log 2>&1
python tools/ab_kernels.py --workload C3 --option persistent_blocks --values 1024 2048 4096 --frames 60 > $OUT/c3_pblocks_v3.log 2>&1
for e in 4 16; do
  VOXELHASH_LIB=$PWD/voxelhashing_demo_amd/lib/alt/libvoxelhash_hip_epl$e.so python tools/ab_kernels.py --workload C3 --option flatten_variant --values 3 5 --frames 60 > $OUT/c3_epl$e.log 2>&1
  VOXELHASH_LIB=$PWD/voxelhashing_demo_amd/lib/alt/libvoxelhash_hip_epl$e.so python tools/ab_kernels.py --workload C2 --option flatten_variant --values 3 5 --frames 60 > $OUT/c2_epl$e.log 2>&1
done
python tools/ab_kernels.py --workload C2 --option flatten_variant --values 3 5 --frames 60 > $OUT/c2_variant.log 2>&1
python tools/ab_kernels.py --workload C3 --option integrate_grid --values 1024 2048 4096 8192 --frames 60 > $OUT/c3_igrid.log 2>&1
python tools/ab_kernels.py --workload C3 --option commit_blocks --values 32 128 512 --frames 60 > $OUT/c3_cblocks.log 2>&1
tail -n 4 $OUT/*.log
