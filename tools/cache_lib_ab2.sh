#!/bin/bash
# Developer A/B (GPU box): the cache tier's one launch (tools/cache_bench.py 16384 300 0) through the in-tree library and through
# a variant built by tools/variants.sh (name given as $1), three times each, interleaved.
# usage: tools/cache_lib_ab2.sh <variant name> [label of the in-tree build] [label of the variant]
V=${1:?variant name}
LA=${2:-in-tree}
LB=${3:-$V}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
show() { python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('%-14s %.2f us per batch, hit rate %.4f' % (sys.argv[1], j['ms_per_step']*1e3, j['hit_rate']))" "$1"; }
for i in 1 2 3; do
  python3 tools/cache_bench.py 16384 300 0 2>/dev/null | show "$LA"
  EVS_LIB_PATH=$ROOT/ev-store-dlrm_amd/lib/var/libevstore_hip_$V.so python3 tools/cache_bench.py 16384 300 0 2>/dev/null | show "$LB"
done
