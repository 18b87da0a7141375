#!/bin/bash
# On the GPU box (one gpurun call): round 4's bench lines, the rocprofv3 kernel statistics of the bench command, the HBM-traffic
# and SQ counter passes (separate --pmc runs, never combined with tracing), the sweep and the side benchmarks
# -> gpurun_out/<tag>/.  The summaries worth keeping are copied into profiles/ by hand afterwards.
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python3 bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json
python3 bench.py --steps 20 --warmup 5 2> $OUT/bench_driver20.err | tail -1 > $OUT/bench_driver20.json
python3 bench.py --force-sharded --steps 2000 --warmup 200 2>/dev/null | tail -1 > $OUT/bench_force_sharded.json
python3 bench.py --force-sharded --exchange-mode p2p --steps 2000 --warmup 200 2>/dev/null | tail -1 > $OUT/bench_force_sharded_p2p.json
python3 bench.py --force-sharded --force-exchange --steps 2000 --warmup 200 2>/dev/null | tail -1 > $OUT/bench_force_sharded_exchange.json
python3 bench.py --force-sharded --placement rowsplit --exchange-mode p2p --steps 2000 --warmup 200 2>/dev/null | tail -1 > $OUT/bench_force_sharded_rowsplit_p2p.json
for a in "2048 100 64 8" "2048 100 36 8" "2048 38 36 26 200000" "4096 17 36 26 200000" "2048 100 128 8" "2048 100 32 8" "2048 100 16 8"; do python3 tools/longbag_bench.py $a 2>/dev/null | tail -1; done > $OUT/longbag.log
python3 tools/multihot_bench.py 16384 10 2>/dev/null | tail -1 > $OUT/multihot.log
python3 tools/multihot_bench.py 4096 40 2>/dev/null | tail -1 >> $OUT/multihot.log
python3 tools/multihot_bench.py 2048 100 2>/dev/null | tail -1 >> $OUT/multihot.log
for b in 16 8 4; do python3 tools/kbench.py --bits $b --codes encoded --batch 16384 65536 2>/dev/null | grep "gather only"; done > $OUT/kbench_reduced.log
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 2000 --warmup 500 --no-cpu-baseline --no-extras > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); grep -E "evs::|Name" $f > $OUT/kernel_stats.csv
rm -rf $OUT/trace
timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --output-format csv -d $OUT/pmc_rd -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-cache-tier > $OUT/pmc_rd.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/pmc_wr -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-cache-tier > $OUT/pmc_wr.log 2>&1
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM" \
  "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS" ; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --output-format csv -d $OUT/sq$i -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-cache-tier > $OUT/sq$i.log 2>&1 || echo "sq pass $i failed"
done
python3 $ROOT/tools/pmc_summary.py $OUT "emb_interact_rf_kernel" > $OUT/pmc_summary.txt
# the reference's benchmark shape: kernel statistics + traffic of the long-bag kernel
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lbtrace -- python3 $ROOT/tools/longbag_bench.py 2048 100 64 8 > $OUT/longbag_prof.log 2>&1
f=$(find $OUT/lbtrace -name "*kernel_stats.csv" | head -1); grep -E "evs::|Name" $f > $OUT/longbag_kernel_stats.csv
rm -rf $OUT/lbtrace
timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --kernel-include-regex "evs::bag_sum_long" --output-format csv -d $OUT/lbpmc/rd -- python3 $ROOT/tools/longbag_bench.py 2048 100 64 8 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_EA0_RDREQ_128B_sum --kernel-include-regex "evs::bag_sum_long" --output-format csv -d $OUT/lbpmc/wr -- python3 $ROOT/tools/longbag_bench.py 2048 100 64 8 > /dev/null 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/lbpmc "evs::" > $OUT/longbag_pmc_summary.txt
# reduced-precision fused launches
for bits in 16 8 4; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rqtrace -- python3 $ROOT/tools/kbench.py --fused-only --bits $bits --codes encoded --batch 16384 65536 --iters 300 > $OUT/kbench_u$bits.log 2>&1
  f=$(find $OUT/rqtrace -name "*kernel_stats.csv" | head -1); grep -E "evs::|Name" $f > $OUT/u${bits}_kernel_stats.csv
  rm -rf $OUT/rqtrace
done
cd $ROOT
timeout 900 python3 tools/sweep.py > $OUT/sweep.md 2> $OUT/sweep.err
python3 tools/h2d_bench.py > $OUT/h2d.log 2>/dev/null
bash tools/prof_cache_r03.sh $TAG > $OUT/prof_cache.log 2>&1
find $OUT -name "*.csv" -size +3M -delete
find $OUT -name "*.db" -delete
du -sh $OUT | tail -1
cat $OUT/pmc_summary.txt | head -40
