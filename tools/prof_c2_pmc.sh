#!/bin/bash
# On the GPU box: HBM traffic of the two- / three-tier batched lookups (configs[4]) -- two separate --pmc passes over tools/c2bench.py
TAG=${1:-c2_pmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --kernel-include-regex "evs::(interact_mixed84|cache_batch_sampled_list2|cache_batch_sa_list2)" --output-format csv -d $OUT/pmc_rd -- python3 $ROOT/tools/c2bench.py > $OUT/pmc_rd.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_EA0_RDREQ_128B_sum --kernel-include-regex "evs::(interact_mixed84|cache_batch_sampled_list2|cache_batch_sa_list2)" --output-format csv -d $OUT/pmc_wr -- python3 $ROOT/tools/c2bench.py > $OUT/pmc_wr.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT "evs::" > $OUT/c2_pmc_summary.txt
find $OUT -name "*.csv" -size +3M -delete
cat $OUT/c2_pmc_summary.txt
