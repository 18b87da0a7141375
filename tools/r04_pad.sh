#!/bin/bash
# developer A/B: the set records' allocation padded to 0 / 64 / 1024 MB (page-size hypothesis of the round-4 layout: rejected)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for pad in 0 64 1024; do
  echo "== EVS_SA_PAD_MB=$pad"
  EVS_SA_PAD_MB=$pad python3 $R/tools/cache_bench.py 16384 300 0 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  %.2f us/batch  hit %.4f' % (r['ms_per_step']*1e3, r['hit_rate']))"
  EVS_SA_PAD_MB=$pad python3 $R/tools/c2bench.py 2>/dev/null | grep -E "mixed-codec"
done
