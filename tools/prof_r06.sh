#!/bin/bash
# On the GPU box (one gpurun call): round 6's evidence on the final tree -> gpurun_out/<tag>/ (the summaries worth keeping are copied
# into profiles/ as r06_*).  Bench lines (default flags, the driver's flags, the N > 1 code path on one rank through the
# self-launcher: direct / inline / p2p exchange forced, no exchange), kernel statistics of the bench command, the HBM-traffic
# passes of the headline launch, the pair (configs[4]: kernel statistics + deciles, SQ counters, HBM traffic over
# tools/c2_pair_bench.py), the cache tier (configs[2]: tools/prof_cache_r05.sh's passes + SQ counters of the PROBE instantiation
# beside the plain launch's), the resident dispatcher and the exact engine's server.  Every --pmc pass is a run of its own.
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python3 bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/bench_driver20.err | tail -1 > $OUT/bench_driver20.json
for mode in direct inline p2p; do
  timeout 300 python3 bench.py --gpus 1 --self-launch --force-sharded --force-exchange --exchange-mode $mode --steps 2000 --warmup 200 2>/dev/null | tail -1 > $OUT/bench_force_sharded_exchange_$mode.json
done
timeout 300 python3 bench.py --gpus 1 --self-launch --force-sharded --steps 2000 --warmup 200 2>/dev/null | tail -1 > $OUT/bench_force_sharded.json
timeout 300 python3 tools/serve_bench.py 1 128 2048 16384 2>&1 | grep "^B=" > $OUT/serve_bench.log
python3 tools/b1_serve_bench.py 2>/dev/null | tail -4 > $OUT/b1_serve.log
python3 tools/exact_replay_bench.py 2>/dev/null | tail -2 >> $OUT/b1_serve.log
for i in 1 2; do python3 tools/c2_pair_bench.py 200 2>&1 | grep "tier batched"; done > $OUT/c2_pair_bench.log
python3 tools/c2_pair_bench.py 200 3 2>&1 | grep "tier batched" >> $OUT/c2_pair_bench.log
cd /tmp && export TMPDIR=/tmp
# ---- the headline: kernel statistics, HBM traffic
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 2000 --warmup 500 --no-cpu-baseline --no-extras --no-cache-tier > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); grep -E "evs::|Name" $f > $OUT/kernel_stats.csv
rm -rf $OUT/trace
timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --output-format csv -d $OUT/pmc/rd -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-cache-tier > $OUT/pmc_rd.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/pmc/wr -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-cache-tier > $OUT/pmc_wr.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/pmc "emb_interact_rf_kernel" > $OUT/pmc_summary.txt
# ---- the sharded step with the direct exchange: which kernels a step is
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/shtrace -- python3 $ROOT/bench.py --gpus 1 --force-sharded --force-exchange --steps 2000 --warmup 100 > $OUT/shtrace.log 2>&1
f=$(find $OUT/shtrace -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-260 > $OUT/sharded_direct_kernel_stats.csv
rm -rf $OUT/shtrace
# ---- the resident dispatcher: the grid's launches (one per idle period) and nothing per batch
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/srvtrace -- python3 $ROOT/tools/serve_bench.py 16384 > $OUT/srvtrace.log 2>&1
f=$(find $OUT/srvtrace -name "*kernel_stats.csv" | head -1); grep -E "evs::|Name" $f | cut -c1-260 > $OUT/serve_kernel_stats.csv
rm -rf $OUT/srvtrace
# ---- the pair: kernel statistics and deciles, SQ counters, HBM traffic
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2trace -- python3 $ROOT/tools/c2_pair_bench.py 200 > $OUT/c2trace.log 2>&1
f=$(find $OUT/c2trace -name "*kernel_stats.csv" | head -1); grep -E "evs::|Name" $f > $OUT/c2_kernel_stats.csv
t=$(find $OUT/c2trace -name "*kernel_trace.csv" | head -1); python3 $ROOT/tools/ktrace_deciles.py $t | grep "evs::" > $OUT/c2_kernel_deciles.txt
rm -rf $OUT/c2trace
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM" \
  "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-include-regex "evs::(interact_mixed84|cache_batch_sa_list2)" --output-format csv -d $OUT/c2sq/pass$i -- python3 $ROOT/tools/c2_pair_bench.py 60 > $OUT/c2sq_pass$i.log 2>&1 || echo "c2 sq pass $i failed"
  timeout 240 rocprofv3 --pmc $set --kernel-include-regex "evs::emb_interact_rf_kernel" --output-format csv -d $OUT/cachesq/pass$i -- python3 $ROOT/tools/cache_bench.py 16384 100 0 > $OUT/cachesq_pass$i.log 2>&1 || echo "cache sq pass $i failed"
  timeout 240 rocprofv3 --pmc $set --kernel-include-regex "evs::emb_interact_rf_kernel" --output-format csv -d $OUT/plainsq/pass$i -- python3 $ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras --no-cache-tier > $OUT/plainsq_pass$i.log 2>&1 || echo "plain sq pass $i failed"
done
python3 $ROOT/tools/pmc_summary.py $OUT/c2sq "evs::" > $OUT/c2_sq_summary.txt
( echo "== cache tier (tools/cache_bench.py 16384 100 0): the PROBE instantiation"; python3 $ROOT/tools/pmc_summary.py $OUT/cachesq "evs::";
  echo "== plain launch (bench.py --no-extras --no-cache-tier)"; python3 $ROOT/tools/pmc_summary.py $OUT/plainsq "evs::" ) > $OUT/cache_sq_summary.txt
timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --kernel-include-regex "evs::(interact_mixed84|cache_batch_sa_list2)" --output-format csv -d $OUT/c2pmc/rd -- python3 $ROOT/tools/c2_pair_bench.py 60 > $OUT/c2pmc_rd.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_EA0_RDREQ_128B_sum --kernel-include-regex "evs::(interact_mixed84|cache_batch_sa_list2)" --output-format csv -d $OUT/c2pmc/wr -- python3 $ROOT/tools/c2_pair_bench.py 60 > $OUT/c2pmc_wr.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/c2pmc "evs::" > $OUT/c2_pmc_summary.txt
# ---- the cache tier: kernel statistics, deciles, HBM traffic (round 5's script)
cd $ROOT
bash tools/prof_cache_r05.sh $TAG/cache > /dev/null 2>&1
cp $OUT/cache/cache_kernel_stats.csv $OUT/cache/cache_kernel_deciles.txt $OUT/cache/cache_pmc_summary.txt $OUT/cache/cache_bench_600.json $OUT/cache/cache_bench_600_noprof.json $OUT/ 2>/dev/null
find $OUT -name "*.csv" -size +3M -delete
find $OUT -name "*.db" -delete
rm -rf $OUT/pmc $OUT/c2sq $OUT/cachesq $OUT/plainsq $OUT/c2pmc $OUT/cache/pmc
du -sh $OUT | tail -1
cat $OUT/pmc_summary.txt | head -12; head -4 $OUT/kernel_stats.csv | cut -c1-200; cat $OUT/serve_bench.log $OUT/b1_serve.log $OUT/c2_pair_bench.log; cat $OUT/sharded_direct_kernel_stats.csv $OUT/serve_kernel_stats.csv | cut -c1-200
