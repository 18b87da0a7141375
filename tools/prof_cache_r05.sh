#!/bin/bash
# On the GPU box, round 5: the cache tier (configs[2]) with the policy update inside the probe launch -- kernel statistics and
# per-dispatch deciles over 600 unseen batches, the two-launch chain beside it (EVS_CACHE_INLINE=0), then the HBM-traffic PMC
# passes (tools/prof_cache_pmc.sh).  -> gpurun_out/<tag>/
TAG=${1:-cache_r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ctrace -- python3 $ROOT/tools/cache_bench.py 16384 600 0 > $OUT/cache_bench_600.json 2> $OUT/cache_bench_600.err
f=$(find $OUT/ctrace -name "*kernel_stats.csv" | head -1); grep -E "evs::|Name" $f > $OUT/cache_kernel_stats.csv
t=$(find $OUT/ctrace -name "*kernel_trace.csv" | head -1); python3 $ROOT/tools/ktrace_deciles.py $t | grep "evs::" > $OUT/cache_kernel_deciles.txt
rm -rf $OUT/ctrace
cd $ROOT
python3 tools/cache_bench.py 16384 600 0 > $OUT/cache_bench_600_noprof.json 2>/dev/null
EVS_CACHE_INLINE=0 python3 tools/cache_bench.py 16384 600 0 > $OUT/cache_bench_600_two_launches.json 2>/dev/null
bash tools/prof_cache_pmc.sh $TAG/pmc > /dev/null 2>&1
cp $OUT/pmc/cache_pmc_summary.txt $OUT/cache_pmc_summary.txt
cat $OUT/cache_kernel_stats.csv $OUT/cache_kernel_deciles.txt $OUT/cache_pmc_summary.txt
for f in $OUT/cache_bench_600*.json; do echo $f; cut -c1-200 $f; done
