#!/bin/bash
# Developer A/B on the GPU box: time the fused kernel of each variant built by tools/variants.sh
for v in ${VARS:-base}; do
  echo "== $v"
  EVS_LIB_PATH=$PWD/ev-store-dlrm_amd/lib/var/libevstore_hip_$v.so timeout 120 python tools/kbench.py --fused-only --batch ${BATCHES:-16384 65536} --iters 300 2>&1 | grep "one index"
done
