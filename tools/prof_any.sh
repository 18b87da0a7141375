#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of any python script of this repo: tools/prof_any.sh <tag> <script.py> [args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
S=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $S "$@" > $OUT/trace.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT -name "*.csv" -size +3M -delete
grep -v rocprofv3 $OUT/trace.log | tail -3; cut -d, -f1-4 $OUT/kernel_stats.csv | head -14 | cut -c1-150
