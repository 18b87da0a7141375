#!/bin/bash
# On the GPU box: per-kernel times of the batched cache tier (tools/cbench.py) -> gpurun_out/<tag>/
TAG=${1:-cprof}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/cbench.py "$@" > $OUT/trace.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT -name "*.csv" -size +3M -delete
tail -2 $OUT/trace.log; cut -d, -f1-4 $OUT/kernel_stats.csv | grep -i "cache\|lds_kernel" | cut -c1-140
