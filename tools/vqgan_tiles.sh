#!/bin/bash
# decode / encode time of the 3D-VQGAN per forced convolution tile (MEBT_CONV_TILE=bm,bn,ring); GPU box
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
for t in "" 128,128,2 128,128,3 256,128,2 128,64,2 128,64,3 256,64,2 256,64,3; do
  echo "tile [$t]: $(MEBT_CONV_TILE=$t timeout 120 python3 tools/vqgan_bench.py 4 2>&1 | grep f16)"
done
