#!/bin/bash
# On the GPU box: per-dispatch duration deciles of the cache tier's kernels (300 unseen batches at capacity) with the
# update inside the probe launch on / off (EVS_CACHE_INLINE).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for f in ${VALS:-1 0}; do
  export EVS_CACHE_INLINE=$f
  rm -rf $R/gpurun_out/dtrace
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/dtrace -- python3 $R/tools/cache_bench.py 16384 300 0 > $R/gpurun_out/dtrace_$f.log 2>&1
  t=$(find $R/gpurun_out/dtrace -name "*kernel_trace.csv" | head -1)
  echo "== EVS_CACHE_INLINE=$f"; python3 $R/tools/ktrace_deciles.py $t | grep "evs::"
  rm -rf $R/gpurun_out/dtrace
done
