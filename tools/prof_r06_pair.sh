#!/bin/bash
# On the GPU box: the two-tier pair after a kernel change -- the pair's tests, tools/c2_pair_bench.py (events), one SQ pass (VALU / SALU /
# LDS conflicts of the consumer) -> stdout
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r06pair}
mkdir -p $OUT
cd $ROOT
timeout 900 python3 -m pytest tests/test_gpu_cache.py tests/test_gpu_tiers_host.py -x -q -m gpu -k "c1c2 or pair or tier or mixed or two" 2>&1 | tail -3 | cut -c1-300
for i in 1 2; do python3 tools/c2_pair_bench.py 200 2>&1 | grep "tier batched"; done
python3 tools/c2_pair_bench.py 200 3 2>&1 | grep "tier batched"
cd /tmp && export TMPDIR=/tmp
i=0
for set in \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" ; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-include-regex "evs::(interact_mixed84|cache_batch_sa_list2)" --output-format csv -d $OUT/sq/pass$i -- python3 $ROOT/tools/c2_pair_bench.py 40 > $OUT/sq_pass$i.log 2>&1 || echo "sq pass $i failed"
done
python3 $ROOT/tools/pmc_summary.py $OUT/sq "evs::interact_mixed84" | tee $OUT/pair_sq_summary.txt
find $OUT -name "*.csv" -size +3M -delete; find $OUT -name "*.db" -delete
