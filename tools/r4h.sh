#!/bin/bash
OUT=gpurun_out/r4h; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_vqgan.py -m gpu -x -q -s > $OUT/tests_vq.log 2>&1; echo "vq rc $?" >> $OUT/tests_vq.log; grep -E "vqgan|passed|failed|rc " $OUT/tests_vq.log | tail -12
python tools/vqgan_bench.py 16 f16 2>&1 | tail -1; MEBT_CONV_THIN_MFMA=0 python tools/vqgan_bench.py 16 f16 2>&1 | tail -1
python tools/vqgan_bench.py 4 f16 2>&1 | tail -1
