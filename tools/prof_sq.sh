#!/bin/bash
# usage: tools/prof_sq.sh <tag> <python args...> : two SQ counter passes, summary printed
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/p1 -- python3 "$@" > $OUT/p1.log 2>&1
timeout 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d $OUT/p2 -- python3 "$@" > $OUT/p2.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT emb_interact
find $OUT -name "*.csv" -size +2M -delete
