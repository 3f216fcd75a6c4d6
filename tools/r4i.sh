#!/bin/bash
OUT=gpurun_out/r4i; mkdir -p $OUT
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/stv -o b -- python3 $OLDPWD/tools/vqgan_bench.py 16 f16 > $OLDPWD/$OUT/vqgan16.txt 2>&1)
cp $OUT/stv/*kernel_stats.csv $OUT/vqgan16_kernel_stats.csv; rm -rf $OUT/stv; python tools/kernel_table.py $OUT/vqgan16_kernel_stats.csv 12
