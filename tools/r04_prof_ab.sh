#!/bin/bash
# On the GPU box: per-kernel statistics of tools/c2bench.py and tools/cache_bench.py under two libraries (A/B by EVS_LIB_PATH)
R=${GRAFT_REPO_ROOT:-$(pwd)}
BASE=$(readlink -f $1)
NEW=$R/ev-store-dlrm_amd/lib/libevstore_hip.so
cd /tmp && export TMPDIR=/tmp
for side in base new; do
  L=$BASE; [ $side = new ] && L=$NEW
  export EVS_LIB_PATH=$L
  for prog in "c2bench.py" "cache_bench.py 16384 200 0"; do
    tag=$(echo $prog | cut -d. -f1)
    rm -rf /tmp/prof_$side_$tag
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${side}_$tag -- python3 $R/tools/$prog > /tmp/prof_${side}_$tag.log 2>&1
    f=$(find /tmp/prof_${side}_$tag -name "*kernel_stats.csv" | head -1)
    echo "== $side $prog"
    python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "evs::" in r["Name"]:
        print("  %-110s calls %6s avg %9.1f ns  min %9s" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]), r["MinNs"]))
PY
  done
done
