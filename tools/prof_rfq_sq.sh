#!/bin/bash
# On the GPU box: SQ issue / wait / matrix-pipe counters of the reduced-precision fused launch (evs_fused_rfq.hip); separate
# --pmc passes over kbench, never combined with tracing -> gpurun_out/<tag>/rfq_sq_summary_u<bits>_B<batch>.txt
TAG=${1:-rfq_sq}; BITS=${2:-8}; BATCH=${3:-65536}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM" \
  "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
  "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-include-regex "emb_interact_rfq" --output-format csv -d $OUT/pass$i -- python3 $ROOT/tools/kbench.py --fused-only --bits $BITS --codes encoded --batch $BATCH --iters 60 > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 $ROOT/tools/pmc_summary.py $OUT "emb_interact_rfq" > $OUT/rfq_sq_summary_u${BITS}_B${BATCH}.txt
find $OUT -name "*.csv" -size +3M -delete
cat $OUT/rfq_sq_summary_u${BITS}_B${BATCH}.txt
