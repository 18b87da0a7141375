#!/bin/bash
# Developer A/B on the GPU box: exact-kernel duration of each variant (tools/variants.sh) for batch-1 requests
R=$PWD
for v in ${VARS:-base}; do
  echo "== $v"
  export EVS_LIB_PATH=$R/ev-store-dlrm_amd/lib/var/libevstore_hip_$v.so
  (cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abx_$v -- python3 $R/tools/b1bench.py > $R/gpurun_out/abx_$v.log 2>&1)
  find $R/gpurun_out/abx_$v -name "*kernel_stats.csv" | head -1 | xargs -I{} cut -d, -f1-4,6,7 {} | grep -i "exact" | cut -c1-160
  find $R/gpurun_out/abx_$v -name "*.csv" -size +1M -delete
done
