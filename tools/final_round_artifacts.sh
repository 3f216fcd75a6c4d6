#!/bin/bash
# End-of-round artifacts on one GPU box: the data-parallel path's bench lines (one-rank RCCL group; two ranks sharing the GPU over gloo),
# the default bench line, and the whole GPU suite.   tools/final_round_artifacts.sh <tag>
TAG=${1:-r05}
OUT=gpurun_out/$TAG; mkdir -p $OUT
MEBT_DP_FORCE=1 python bench.py --secondary none --no-cpu-baseline > $OUT/bench_dp_path_one_rank_rccl.json 2> $OUT/bench_dp_one.err; head -c 400 $OUT/bench_dp_path_one_rank_rccl.json; echo
MEBT_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 8 --warmup 3 --secondary none --no-cpu-baseline > $OUT/bench_share_gpu_2ranks_gloo.json 2> $OUT/bench_dp_two.err; head -c 400 $OUT/bench_share_gpu_2ranks_gloo.json; echo
python bench.py > $OUT/bench_n1_full.json 2> $OUT/bench_n1_full.err; head -c 700 $OUT/bench_n1_full.json; echo
timeout 2700 python -m pytest tests -m gpu -q --durations=12 > $OUT/gpu_suite.log 2>&1; echo "suite rc $?" >> $OUT/gpu_suite.log; tail -22 $OUT/gpu_suite.log
