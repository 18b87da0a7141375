#!/bin/bash
# On the GPU box: the batched cache tier with the in-batch fork off / on -- per-batch time over 300 unseen batches and
# the bench form (30 batches after a replay settle), then the per-kernel duration deciles from a kernel trace.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for f in ${FORKS:-0 1}; do
  echo "== EVS_CACHE_FORK=$f"
  EVS_CACHE_FORK=$f python3 $R/tools/cache_bench.py 16384 300 0 2>/dev/null | cut -c1-120
  EVS_CACHE_FORK=$f python3 $R/tools/cache_bench.py 16384 30 0.35 2>/dev/null | cut -c1-120
done
cd /tmp && export TMPDIR=/tmp
for f in ${FORKS:-0 1}; do
  echo "== kernel durations (us), deciles, EVS_CACHE_FORK=$f"
  EVS_CACHE_FORK=$f timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ctrace2 -- python3 $R/tools/cache_bench.py 16384 200 0 > $R/gpurun_out/ctrace2.log 2>&1
  t=$(find $R/gpurun_out/ctrace2 -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/ktrace_deciles.py $t | grep evs::
  rm -rf $R/gpurun_out/ctrace2
done
