#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for f in ${FLAGS:-0}; do
  echo "== EVS_DBG_FLAGS=$f"
  EVS_DBG_FLAGS=$f timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ctrace2 -- python3 $R/tools/cache_bench.py 16384 200 0 > $R/gpurun_out/ctrace2.log 2>&1
  t=$(find $R/gpurun_out/ctrace2 -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/ktrace_deciles.py $t | grep "evs::cache_batch_sampled_kernel\|probe_gather"
  rm -rf $R/gpurun_out/ctrace2
done
