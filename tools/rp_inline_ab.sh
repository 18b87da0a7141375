#!/bin/bash
# On the GPU box: the single reduced-precision set-associative tier (tools/cache_rp_bench.py) with the policy update inside the
# probe + interaction launch (EVS_CACHE_INLINE=1) against the two-launch chain, and library variants (lib/var) beside them.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
  for f in 0 1; do
    echo "== EVS_CACHE_INLINE=$f"
    EVS_CACHE_INLINE=$f python3 $R/tools/cache_rp_bench.py ${BITS:-8 4 16} 2>/dev/null
  done
  for v in ${VARIANTS:-}; do
    echo "== variant $v (inline)"
    EVS_LIB_PATH=$R/ev-store-dlrm_amd/lib/var/libevstore_hip_$v.so python3 $R/tools/cache_rp_bench.py ${BITS:-8 4 16} 2>/dev/null
  done
done
