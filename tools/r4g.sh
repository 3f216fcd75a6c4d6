#!/bin/bash
OUT=gpurun_out/r4g; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_dropout.py -m gpu -x -q -k "attention or dropout" > $OUT/tests_attn.log 2>&1; echo "attn rc $?" >> $OUT/tests_attn.log; tail -4 $OUT/tests_attn.log
python tools/attn_bench.py > $OUT/attn.txt 2>&1; grep -v amdgpu $OUT/attn.txt
timeout 900 python -m pytest tests/test_gpu_benchsize.py -m gpu -x -q -k "c2_bf16_train or c4_train" > $OUT/tests_bs.log 2>&1; echo "bs rc $?" >> $OUT/tests_bs.log; tail -4 $OUT/tests_bs.log
python bench.py --secondary light --no-cpu-baseline > $OUT/bench_head.json 2> $OUT/bench_head.err; head -c 900 $OUT/bench_head.json; echo
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/stv -o b -- python3 $OLDPWD/tools/vqgan_bench.py 16 f16 > $OLDPWD/$OUT/vqgan16.txt 2>&1)
cp $OUT/stv/*kernel_stats.csv $OUT/vqgan16_kernel_stats.csv; rm -rf $OUT/stv; python tools/kernel_table.py $OUT/vqgan16_kernel_stats.csv 30 | tee $OUT/vqgan16_table.txt; cat $OUT/vqgan16.txt | tail -2
