#!/bin/bash
# final artifacts, part 1: kernel statistics + PMC passes of the headline step, config-4 / config-5 kernel statistics, micro-benchmarks
bash tools/profile_round.sh r04 > gpurun_out/r04_profile_round.log 2>&1; tail -8 gpurun_out/r04_profile_round.log
python tools/pmc_sq.py 2>/dev/null | head -1
bash tools/profile_c45.sh r04_c45 > gpurun_out/r04_c45.log 2>&1; tail -3 gpurun_out/r04_c45.log
OUT=gpurun_out/r04_micro; mkdir -p $OUT
for W in 8 4; do MEBT_ATTN_FWD_WAVES=$W python tools/attn_bench.py > $OUT/attn_w$W.txt 2>&1; done; python tools/attn_bench.py > $OUT/attn_auto.txt 2>&1
python tools/sample_kernel_bench.py > $OUT/sample_fast.txt 2>&1; MEBT_SAMPLE_FAST=0 python tools/sample_kernel_bench.py > $OUT/sample_old.txt 2>&1
python tools/vqgan_bench.py 16 f16 > $OUT/vqgan16.txt 2>&1; MEBT_CONV_THIN_MFMA=0 MEBT_CODEBOOK_FILTER=0 python tools/vqgan_bench.py 16 f16 > $OUT/vqgan16_r03_kernels.txt 2>&1
python tools/gemm_bench.py --torch > $OUT/gemm_vs_hipblaslt_warm.txt 2>&1; python tools/gemm_bench.py --torch --cold > $OUT/gemm_vs_hipblaslt_cold.txt 2>&1
tail -3 $OUT/gemm_vs_hipblaslt_warm.txt
