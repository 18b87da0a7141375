#!/bin/bash
# usage (on the GPU box): tools/prof_pmc.sh <outdir> -- <python args...>
# Runs separate rocprofv3 --pmc passes (never combined with tracing, per the pool's rule).
set -u
OUT=$1; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM" \
  "FETCH_SIZE TCC_MISS_sum" \
  "WRITE_SIZE TCC_HIT_sum TCC_EA0_RDREQ_sum" \
  "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_REQ_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum" \
  "TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_BUSY_avr" \
  "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $ROOT/$OUT/pass$i -- python3 "$@" > $ROOT/$OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 $ROOT/tools/pmc_summary.py $ROOT/$OUT | tee $ROOT/$OUT/summary.txt
# keep only the summary (raw csv can be large)
find $ROOT/$OUT -name "*.csv" -size +2M -delete
