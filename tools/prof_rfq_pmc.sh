#!/bin/bash
# On the GPU box: HBM traffic of the reduced-precision fused launch (evs_fused_rfq.hip), separate --pmc passes over kbench
TAG=${1:-rfq_pmc}; BITS=${2:-16}; BATCH=${3:-16384}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --kernel-include-regex "emb_interact_rfq" --output-format csv -d $OUT/pmc_rd -- python3 $ROOT/tools/kbench.py --fused-only --bits $BITS --codes encoded --batch $BATCH --iters 100 > $OUT/pmc_rd.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_EA0_RDREQ_128B_sum --kernel-include-regex "emb_interact_rfq" --output-format csv -d $OUT/pmc_wr -- python3 $ROOT/tools/kbench.py --fused-only --bits $BITS --codes encoded --batch $BATCH --iters 100 > $OUT/pmc_wr.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT "emb_interact_rfq" > $OUT/rfq_pmc_summary_u${BITS}_B${BATCH}.txt
find $OUT -name "*.csv" -size +3M -delete
cat $OUT/rfq_pmc_summary_u${BITS}_B${BATCH}.txt
