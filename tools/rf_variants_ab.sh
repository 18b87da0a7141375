#!/bin/bash
# Developer A/B (GPU box): variants of evs_fused_rf.hip built by tools/variants.sh (name@evs_fused_rf:"-D..."), timed back to
# back on one box -- boxes differ by +-3 %, so only same-box comparisons count.
# usage: VARS="a b" BATCHES="16384" REPS=3 NB=64 tools/rf_variants_ab.sh   (NB: distinct batches cycled; bench.py uses 64)
for r in $(seq 1 ${REPS:-2}); do
for v in ${VARS:-base}; do
  echo "== $v"
  EVS_LIB_PATH=$PWD/ev-store-dlrm_amd/lib/var/libevstore_hip_$v.so timeout 120 python tools/kbench.py --fused-only --n-batches ${NB:-8} --batch ${BATCHES:-16384} --iters 500 2>&1 | grep "us"
done
done
