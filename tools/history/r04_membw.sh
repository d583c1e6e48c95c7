#!/bin/bash
# The HBM-ceiling probe at three sizes: 419 MB (C3's table, 1.56 x the Infinity Cache),
# 1 680 MB (C5's unsharded table) and 4 000 MB; temporal and non-temporal loads.  Output: gpurun_out/r04/membw.txt
set -u
OUT=gpurun_out/r04; mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/membw tools/micro/membw.hip || exit 1
: > $OUT/membw.txt
for MB in 105 419 1680 4000; do
  echo "== $MB MB" >> $OUT/membw.txt
  /tmp/membw $MB >> $OUT/membw.txt 2>&1
done
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -iE "TCC_EA0?_RD|DRAM|HBM|TCC_REQ|TCC_HIT|TCC_MISS|FETCH_SIZE|MALL" | head -80 ) > $OUT/pmc_avail_dram.txt
cat $OUT/membw.txt
