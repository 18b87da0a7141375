#!/bin/bash
# Developer A/B on the GPU box: per-kernel cache-tier times of each variant (tools/variants.sh)
for v in ${VARS:-base}; do
  echo "== $v"
  export EVS_LIB_PATH=$PWD/ev-store-dlrm_amd/lib/var/libevstore_hip_$v.so
  bash tools/prof_cache.sh cprof_$v ${ALPHA:-0.75} 2>&1 | grep "cache_batch"
done
