#!/bin/bash
# A/B of MEBT_KV_AHEAD (K / V of the 'latent_enc' blocks ahead of the latent chain on the side stream) on one box:
#   config 4 legs (revise, sample, bootstrap, train), config 5, and the config-2 headline.   tools/kv_ahead_ab.sh [rounds]
R=${1:-2}
OUT=gpurun_out/kv_ahead; mkdir -p $OUT
for r in $(seq 1 $R); do
  for mode in 0 -1 1; do
    MEBT_KV_AHEAD=$mode python bench.py --secondary c4 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())['secondary']
print('c4  mode $mode:', ' '.join(f\"{k.replace('c4_', '')} {v.get('s', v.get('ms_per_step'))}\" for k, v in d.items() if k.startswith('c4_')))"
  done
done | tee $OUT/c4.txt
for r in $(seq 1 $R); do
  for mode in 0 -1 1; do
    MEBT_KV_AHEAD=$mode python bench.py --secondary c5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())['secondary']['c5_taichi_end_to_end']
print('c5  mode $mode:', d['sample_64_steps_ms'], d['revise_8x2_ms'], d['total_ms'])"
  done
done | tee $OUT/c5.txt
for r in $(seq 1 $R); do
  for mode in 0 1; do
    MEBT_KV_AHEAD=$mode python bench.py --secondary none --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('c2  mode $mode:', d['ms_per_step'], d['roofline']['gemm_ms_per_step'])"
  done
done | tee $OUT/c2.txt
