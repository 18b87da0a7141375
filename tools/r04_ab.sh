#!/bin/bash
# On the GPU box: A/B by library of the cache tier (configs[2]) and the two-/three-tier chain (configs[4]).
# usage: tools/r04_ab.sh <baseline.so> [rounds]   (the current tree's library is the other side; both through EVS_LIB_PATH = ctypes path)
R=${GRAFT_REPO_ROOT:-$(pwd)}
BASE=$1; N=${2:-2}
NEW=$R/ev-store-dlrm_amd/lib/libevstore_hip.so
for i in $(seq $N); do
  for side in base new; do
    L=$BASE; [ $side = new ] && L=$NEW
    echo "== $side cache_bench 16384 x 300"
    EVS_LIB_PATH=$L python3 $R/tools/cache_bench.py 16384 300 0 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  %.2f us/batch  hit %.4f  frac %.3f' % (r['ms_per_step']*1e3, r['hit_rate'], r['roofline']['frac']))"
    echo "== $side c2bench"
    EVS_LIB_PATH=$L python3 $R/tools/c2bench.py 2>/dev/null | grep -E "mixed-codec|rows out"
  done
done
