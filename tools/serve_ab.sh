#!/bin/bash
# Developer A/B (GPU box): variants of the resident dispatcher (tools/variants.sh name@evs_fused_rf:"-DEVS_X_SRV=..") through tools/serve_bench.py
for v in ${VARS:-base}; do
  echo "== $v ${N_BLOCKS:+n_blocks=$N_BLOCKS}"
  if [ "$v" = base ]; then timeout 200 python3 tools/serve_bench.py ${BATCHES:-1 2048 16384} 2>&1 | grep -E "^B=|fault" | cut -c1-260
  else EVS_LIB_PATH=$PWD/ev-store-dlrm_amd/lib/var/libevstore_hip_$v.so timeout 200 python3 tools/serve_bench.py ${BATCHES:-1 2048 16384} 2>&1 | grep -E "^B=|fault" | cut -c1-260; fi
done
