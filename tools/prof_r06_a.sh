#!/bin/bash
# On the GPU box (one gpurun call), round 6 first pass: the GPU suite, the driver-form bench line, and the evidence the
# previous verdict asked for -- configs[4] pair: kernel statistics + deciles, SQ issue / wait / LDS counters, HBM traffic
# (tools/c2_pair_bench.py: 200 unseen batches at capacity); configs[2]: SQ counters of the PROBE instantiation of the fused
# launch beside the plain launch's.  Every --pmc pass is a run of its own, never combined with tracing.  -> gpurun_out/r06a/
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06a
mkdir -p $OUT
cd $ROOT
timeout 1500 python3 -m pytest tests -x -q -m gpu > $OUT/gputest.log 2>&1; echo "pytest rc $?" >> $OUT/gputest.log
tail -5 $OUT/gputest.log
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver20.json 2> $OUT/bench_driver20.err; echo "bench rc $?"
tail -c 600 $OUT/bench_driver20.json
cd /tmp && export TMPDIR=/tmp
# ---- the pair: kernel statistics and deciles
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2trace -- python3 $ROOT/tools/c2_pair_bench.py 200 > $OUT/c2_pair_bench.log 2>&1
f=$(find $OUT/c2trace -name "*kernel_stats.csv" | head -1); grep -E "evs::|Name" $f > $OUT/c2_kernel_stats.csv
t=$(find $OUT/c2trace -name "*kernel_trace.csv" | head -1); python3 $ROOT/tools/ktrace_deciles.py $t | grep "evs::" > $OUT/c2_kernel_deciles.txt
rm -rf $OUT/c2trace
python3 $ROOT/tools/c2_pair_bench.py 200 > $OUT/c2_pair_bench_noprof.log 2>&1
# ---- the pair: SQ counters
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM" \
  "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-include-regex "evs::(interact_mixed84|cache_batch_sa_list2)" --output-format csv -d $OUT/c2sq/pass$i -- python3 $ROOT/tools/c2_pair_bench.py 60 > $OUT/c2sq_pass$i.log 2>&1 || echo "c2 sq pass $i failed"
done
python3 $ROOT/tools/pmc_summary.py $OUT/c2sq "evs::" > $OUT/c2_sq_summary.txt
# ---- the pair: HBM traffic
timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --kernel-include-regex "evs::(interact_mixed84|cache_batch_sa_list2)" --output-format csv -d $OUT/c2pmc/rd -- python3 $ROOT/tools/c2_pair_bench.py 60 > $OUT/c2pmc_rd.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_EA0_RDREQ_128B_sum --kernel-include-regex "evs::(interact_mixed84|cache_batch_sa_list2)" --output-format csv -d $OUT/c2pmc/wr -- python3 $ROOT/tools/c2_pair_bench.py 60 > $OUT/c2pmc_wr.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/c2pmc "evs::" > $OUT/c2_pmc_summary.txt
# ---- configs[2]: SQ counters of emb_interact_rf_kernel<..., PROBE> (cache_bench) and of the plain launch (bench --no-extras)
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM" \
  "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-include-regex "evs::emb_interact_rf_kernel" --output-format csv -d $OUT/cachesq/pass$i -- python3 $ROOT/tools/cache_bench.py 16384 100 0 > $OUT/cachesq_pass$i.log 2>&1 || echo "cache sq pass $i failed"
  timeout 240 rocprofv3 --pmc $set --kernel-include-regex "evs::emb_interact_rf_kernel" --output-format csv -d $OUT/plainsq/pass$i -- python3 $ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras --no-cache-tier > $OUT/plainsq_pass$i.log 2>&1 || echo "plain sq pass $i failed"
done
( echo "== cache tier (tools/cache_bench.py 16384 100 0): the PROBE instantiation"; python3 $ROOT/tools/pmc_summary.py $OUT/cachesq "evs::";
  echo "== plain launch (bench.py --no-extras --no-cache-tier)"; python3 $ROOT/tools/pmc_summary.py $OUT/plainsq "evs::" ) > $OUT/cache_sq_summary.txt
find $OUT -name "*.csv" -size +3M -delete
find $OUT -name "*.db" -delete
cat $OUT/c2_pair_bench.log $OUT/c2_pair_bench_noprof.log | grep -v rocprofv3 | tail -4
cat $OUT/c2_kernel_stats.csv | cut -c1-220; cat $OUT/c2_kernel_deciles.txt | cut -c1-220
cat $OUT/c2_sq_summary.txt $OUT/c2_pmc_summary.txt $OUT/cache_sq_summary.txt
