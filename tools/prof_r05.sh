#!/bin/bash
# On the GPU box (one gpurun call): round 5's bench lines, the rocprofv3 kernel statistics of the bench command, the HBM-traffic
# counter passes of the headline launch (separate --pmc runs, never combined with tracing), the u8 rows-in-registers launch
# (kernel statistics + SQ counters: the integer-matrix-pipe form), the forced sharded steps -> gpurun_out/<tag>/.
# The cache tier has its own script (tools/prof_cache_r05.sh).  Summaries worth keeping are copied into profiles/ by hand.
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python3 bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json
python3 bench.py --steps 20 --warmup 5 2> $OUT/bench_driver20.err | tail -1 > $OUT/bench_driver20.json
python3 bench.py --force-sharded --steps 2000 --warmup 200 2>/dev/null | tail -1 > $OUT/bench_force_sharded.json
python3 bench.py --force-sharded --force-exchange --steps 2000 --warmup 200 2>/dev/null | tail -1 > $OUT/bench_force_sharded_exchange.json
python3 bench.py --force-sharded --force-exchange --exchange-mode auto --steps 2000 --warmup 200 2>/dev/null | tail -1 > $OUT/bench_force_sharded_auto.json
for b in 16 8 4; do python3 tools/kbench.py --bits $b --codes encoded --batch 16384 65536 2>/dev/null | grep "fused"; done > $OUT/kbench_reduced.log
python3 tools/b1_serve_bench.py 2>/dev/null | tail -4 > $OUT/b1_serve.log
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 2000 --warmup 500 --no-cpu-baseline --no-extras --no-cache-tier > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); grep -E "evs::|Name" $f > $OUT/kernel_stats.csv
rm -rf $OUT/trace
timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --output-format csv -d $OUT/pmc_rd -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-cache-tier > $OUT/pmc_rd.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/pmc_wr -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-cache-tier > $OUT/pmc_wr.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT "emb_interact_rf_kernel" > $OUT/pmc_summary.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rqtrace -- python3 $ROOT/tools/kbench.py --fused-only --bits 8 --codes encoded --batch 16384 65536 --iters 300 > $OUT/kbench_u8.log 2>&1
f=$(find $OUT/rqtrace -name "*kernel_stats.csv" | head -1); grep -E "evs::|Name" $f > $OUT/u8_kernel_stats.csv
rm -rf $OUT/rqtrace
cd $ROOT
bash tools/prof_rfq_sq.sh $TAG/rfq_sq 8 65536 > /dev/null 2>&1
cp $OUT/rfq_sq/rfq_sq_summary_u8_B65536.txt $OUT/ 2>/dev/null
bash tools/prof_c2_pmc.sh $TAG/c2pmc > /dev/null 2>&1
cp $OUT/c2pmc/c2_pmc_summary.txt $OUT/c2_pmc_summary.txt 2>/dev/null
python3 tools/c2bench.py > $OUT/c2bench.log 2>/dev/null
find $OUT -name "*.csv" -size +3M -delete
find $OUT -name "*.db" -delete
du -sh $OUT | tail -1
cat $OUT/pmc_summary.txt | head -20; cat $OUT/kernel_stats.csv | head -5; cat $OUT/u8_kernel_stats.csv; cat $OUT/kbench_reduced.log $OUT/b1_serve.log $OUT/c2bench.log
