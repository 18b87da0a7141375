#!/bin/bash
# Developer A/B on the GPU box: the fused bag-1 launch with the rows-in-flight kernel on / off
for rf in 0 1; do
  echo "== EVS_FUSED_RF=$rf"
  EVS_FUSED_RF=$rf timeout 300 python tools/kbench.py --fused-only --batch ${BATCHES:-2048 4096 16384 65536 131072} --iters 300 2>&1 | grep "one index"
done
