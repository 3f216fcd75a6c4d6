#!/bin/bash
# Spatial split of the backward (MEBT_CU_SPLIT = CUs per XCD for the chain; the weight gradients + AdamW get the rest), A/B on one box:
#   tools/cu_split_ab.sh 3 0 16 20 24      -> R alternating rounds of fresh-tuned bench runs, one column per setting
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
R=$1; shift
ARGS=""
for n in "$@"; do ARGS="$ARGS MEBT_CU_SPLIT=$n"; done
exec tools/env_ab.sh $R $ARGS
