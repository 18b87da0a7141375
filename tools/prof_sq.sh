#!/bin/bash
# On the GPU box: SQ / MFMA counters of the default bench command (two separate --pmc passes, each under timeout;
# never combined with tracing) -> gpurun_out/<tag>/sq_summary.txt
TAG=${1:-sq}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM" \
  "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS" ; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-cache-tier > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 $ROOT/tools/pmc_summary.py $OUT "lds_kernel<32, 2, 1, 2, false, true, false, true, true, false>" > $OUT/sq_summary.txt
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/sq_summary.txt
