#!/bin/bash
# On the GPU box: SQ issue / wait counters of the two-tier chain's kernels (configs[4]) and, for scale, of the u8 fused launch --
# separate --pmc passes over tools/c2bench.py, never combined with tracing -> gpurun_out/<tag>/c2_sq_summary.txt
TAG=${1:-c2_sq}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM" \
  "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-include-regex "evs::(interact_mixed84|cache_batch_sa_list2)" --output-format csv -d $OUT/pass$i -- python3 $ROOT/tools/c2bench.py > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 $ROOT/tools/pmc_summary.py $OUT "evs::" > $OUT/c2_sq_summary.txt
find $OUT -name "*.csv" -size +3M -delete
cat $OUT/c2_sq_summary.txt
