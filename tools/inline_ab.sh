#!/bin/bash
# On the GPU box: the set-associative cache tier with the policy update inside the probe + interaction launch
# (EVS_CACHE_INLINE=1) against the two-launch chain: per-batch time over 300 unseen batches.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for f in ${VALS:-0 1 0 1}; do
  echo "== ${VAR:-EVS_CACHE_INLINE}=$f B=${B:-16384}"
  env ${VAR:-EVS_CACHE_INLINE}=$f python3 $R/tools/cache_bench.py ${B:-16384} 300 0 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: r[k] for k in ('ms_per_step','hit_rate') if k in r}, r.get('roofline',{}).get('avg_launch_ms'))"
done
