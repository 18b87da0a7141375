#!/bin/bash
# On the GPU box: the cache tier alone under its default (set-associative) policy -- kernel statistics and per-dispatch
# deciles over 600 unseen batches, the sampled / plan policies beside it, then the HBM-traffic PMC passes
# (tools/prof_cache_pmc.sh).  -> gpurun_out/<tag>/
TAG=${1:-cache_r03}
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
EVS_CACHE_POLICY=sampled python3 tools/cache_bench.py 16384 600 0 > $OUT/cache_bench_600_sampled.json 2>/dev/null
EVS_CACHE_POLICY=plan python3 tools/cache_bench.py 16384 600 0 > $OUT/cache_bench_600_plan.json 2>/dev/null
bash tools/prof_cache_pmc.sh $TAG/pmc > /dev/null 2>&1
cp $OUT/pmc/cache_pmc_summary.txt $OUT/cache_pmc_summary.txt
cat $OUT/cache_kernel_stats.csv $OUT/cache_kernel_deciles.txt $OUT/cache_pmc_summary.txt
for f in $OUT/cache_bench_600*.json; do echo $f; cut -c1-200 $f; done
# configs[4]: the two- / three-tier chain (set-associative pair by default): kernel statistics, the script's own lines, PMC traffic
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2trace -- python3 $ROOT/tools/c2bench.py > $OUT/c2bench_prof.log 2>&1
f=$(find $OUT/c2trace -name "*kernel_stats.csv" | head -1); grep -E "evs::|Name" $f > $OUT/c2_kernel_stats.csv
rm -rf $OUT/c2trace
cd $ROOT
python3 tools/c2bench.py > $OUT/c2bench.log 2>/dev/null
EVS_CACHE_POLICY=sampled python3 tools/c2bench.py > $OUT/c2bench_sampled.log 2>/dev/null
bash tools/prof_c2_pmc.sh $TAG/c2pmc > /dev/null 2>&1
cp $OUT/c2pmc/c2_pmc_summary.txt $OUT/c2_pmc_summary.txt
cat $OUT/c2_kernel_stats.csv $OUT/c2bench.log $OUT/c2bench_sampled.log $OUT/c2_pmc_summary.txt
