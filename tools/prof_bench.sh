#!/bin/bash
# On the GPU box: kernel trace + two PMC passes of the default bench command; summaries -> gpurun_out/<tag>/
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 2000 --warmup 500 --no-cpu-baseline --no-extras > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --output-format csv -d $OUT/pmc_rd -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-cache-tier > $OUT/pmc_rd.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/pmc_wr -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-cache-tier > $OUT/pmc_wr.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT -name "*.csv" -size +3M -delete
cat $OUT/pmc_summary.txt; head -4 $OUT/kernel_stats.csv | cut -c1-200; tail -2 $OUT/trace.log | cut -c1-600
