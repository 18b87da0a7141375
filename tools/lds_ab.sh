#!/bin/bash
# On the GPU box: the rows-in-registers kernel with the output staging as it is (packed lower triangle: the ds_write_b32
# of one instruction land on overlapping bank ranges) against a build whose staging stores are lane-linear, i.e. free of
# bank conflicts (wrong R, timing only) -- time, and the LDS conflict counters of both.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
EVS_ATTRIB_RF=1 python3 tools/attrib_exp.py rfbase rfflat
cd /tmp && export TMPDIR=/tmp
for v in rfbase rfflat; do
  EVS_LIB_PATH=$R/ev-store-dlrm_amd/lib/var/libevstore_hip_$v.so timeout 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/lds_$v -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-cache-tier > /dev/null 2>&1
  echo "== $v"; python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/lds_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "emb_interact_rf_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("   %-24s mean %12.1f  n=%d" % (k, sum(v) / len(v), len(v)))
PY
  rm -rf $R/gpurun_out/lds_$v
done
