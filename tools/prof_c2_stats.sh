#!/bin/bash
# On the GPU box: kernel statistics + per-dispatch deciles of the two- / three-tier chain (tools/c2bench.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/c2trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c2trace -- python3 $R/tools/c2bench.py > $R/gpurun_out/c2trace.log 2>&1
t=$(find $R/gpurun_out/c2trace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/ktrace_deciles.py $t | grep "evs::" | cut -c1-200
rm -rf $R/gpurun_out/c2trace
