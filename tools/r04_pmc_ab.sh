#!/bin/bash
# On the GPU box: SQ / TCC counters of the cache tier's two kernels under two libraries (A/B by EVS_LIB_PATH)
R=${GRAFT_REPO_ROOT:-$(pwd)}
BASE=$(readlink -f $1)
NEW=$R/ev-store-dlrm_amd/lib/libevstore_hip.so
PROG=${2:-cache_bench.py 16384 60 0}
cd /tmp && export TMPDIR=/tmp
for side in base new; do
  L=$BASE; [ $side = new ] && L=$NEW
  export EVS_LIB_PATH=$L
  i=0
  for set in \
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
    "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INSTS_FLAT" \
    "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
    "TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_ATOMIC_sum" ; do
    i=$((i+1))
    rm -rf /tmp/pmc_${side}_$i
    timeout 240 rocprofv3 --pmc $set --kernel-include-regex "evs::(emb_interact_rf|cache_batch_sa|interact_mixed84)" --output-format csv -d /tmp/pmc_${side}_$i -- python3 $R/tools/$PROG > /tmp/pmc_${side}_$i.log 2>&1 || echo "$side pass $i failed"
  done
  mkdir -p /tmp/pmc_$side && rm -rf /tmp/pmc_$side/* && mv /tmp/pmc_${side}_? /tmp/pmc_$side/
  echo "==== $side"
  python3 $R/tools/pmc_summary.py /tmp/pmc_$side "evs::"
done
