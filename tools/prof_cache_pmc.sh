#!/bin/bash
# On the GPU box: HBM traffic of the cache tier's launch chain (configs[2]) -- two separate --pmc passes over
# tools/cache_bench.py (never combined with tracing), summarised per kernel -> gpurun_out/<tag>/cache_pmc_summary.txt
TAG=${1:-cache_pmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --kernel-include-regex "evs::" --output-format csv -d $OUT/pmc_rd -- python3 $ROOT/tools/cache_bench.py 16384 200 0 > $OUT/pmc_rd.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_EA0_RDREQ_128B_sum --kernel-include-regex "evs::" --output-format csv -d $OUT/pmc_wr -- python3 $ROOT/tools/cache_bench.py 16384 200 0 > $OUT/pmc_wr.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT "evs::" > $OUT/cache_pmc_summary.txt
find $OUT -name "*.csv" -size +3M -delete
cat $OUT/cache_pmc_summary.txt
