#!/bin/bash
# On the GPU box (one gpurun call): the round's bench lines, the rocprofv3 kernel statistics of the bench command,
# the HBM-traffic and SQ counter passes (separate --pmc runs, never combined with tracing), the sweep and the side
# benchmarks -> gpurun_out/<tag>/.  The summaries worth keeping are copied into profiles/ by hand afterwards.
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver20.json 2> $OUT/bench_driver20.err
python3 bench.py --force-sharded --steps 300 --warmup 50 > $OUT/bench_force_sharded.json 2> $OUT/bench_force_sharded.err
python3 bench.py --force-sharded --placement rowsplit --steps 300 --warmup 50 > $OUT/bench_force_sharded_rowsplit.json 2>/dev/null
python3 tools/call_overhead.py 128 > $OUT/call_overhead_128.log 2>&1
python3 tools/call_overhead.py 2048 > $OUT/call_overhead_2048.log 2>&1
python3 tools/multi_probe.py 8 > $OUT/multi_probe_8.log 2>&1
python3 tools/multi_probe.py 2 > $OUT/multi_probe_2.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 2000 --warmup 500 --no-cpu-baseline --no-extras > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
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
# reduced-precision fused launches (tables encoded from the fp32 ones), kernel statistics
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rqtrace -- python3 $ROOT/tools/kbench.py --fused-only --bits 16 --codes encoded --batch 16384 65536 --iters 300 > $OUT/kbench_u16.log 2>&1
f=$(find $OUT/rqtrace -name "*kernel_stats.csv" | head -1); grep -E "evs::|Name" $f > $OUT/u16_kernel_stats.csv
rm -rf $OUT/rqtrace
for bits in 8 4; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rqtrace -- python3 $ROOT/tools/kbench.py --fused-only --bits $bits --codes encoded --batch 16384 65536 --iters 300 > $OUT/kbench_u$bits.log 2>&1
  f=$(find $OUT/rqtrace -name "*kernel_stats.csv" | head -1); grep -E "evs::|Name" $f > $OUT/u${bits}_kernel_stats.csv
  rm -rf $OUT/rqtrace
done
python3 $ROOT/tools/multihot_bench.py 16384 10 > $OUT/multihot.log 2>/dev/null
python3 $ROOT/tools/multihot_bench.py 4096 40 >> $OUT/multihot.log 2>/dev/null
python3 $ROOT/tools/multihot_bench.py 2048 100 >> $OUT/multihot.log 2>/dev/null
cd $ROOT
timeout 900 python3 tools/sweep.py > $OUT/sweep.md 2> $OUT/sweep.err
# the cache tier alone and the two- / three-tier chains: kernel statistics, deciles, the other policies, PMC traffic
bash tools/prof_cache_r03.sh $TAG > $OUT/prof_cache.log 2>&1
python3 tools/a2a_cost.py 2>/dev/null | grep "us/call" > $OUT/a2a_cost.log
python3 tools/copy_probe.py > $OUT/copy_probe.log 2>&1
python3 bench.py --force-sharded --force-exchange --steps 300 --warmup 50 --no-cpu-baseline --no-cache-tier --no-extras > $OUT/bench_force_sharded_exchange.json 2>/dev/null
find $OUT -name "*.csv" -size +3M -delete
find $OUT -name "*.db" -delete
du -sh $OUT | tail -1
cat $OUT/pmc_summary.txt | head -40
