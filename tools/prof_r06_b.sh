#!/bin/bash
# On the GPU box, round 6 second pass: the rest of the GPU suite (no -x), the sharded step on one rank through the self-launcher
# with the exchange forced -- the extension's own ncclAllToAllv ("direct"), all_to_all_single ("inline"), p2p, no exchange --
# and a kernel trace of the direct step.  -> gpurun_out/r06b/
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06b
mkdir -p $OUT
cd $ROOT
timeout 1700 python3 -m pytest tests -q -m gpu > $OUT/gputest.log 2>&1; echo "pytest rc $?" >> $OUT/gputest.log
tail -8 $OUT/gputest.log
for mode in direct inline p2p; do
  timeout 300 python3 bench.py --gpus 1 --self-launch --force-sharded --force-exchange --exchange-mode $mode --steps 2000 --warmup 100 > $OUT/sharded_exchange_$mode.json 2> $OUT/sharded_exchange_$mode.err
  echo "$mode rc $?"; tail -c 400 $OUT/sharded_exchange_$mode.err
done
timeout 300 python3 bench.py --gpus 1 --self-launch --force-sharded --steps 2000 --warmup 100 > $OUT/sharded_noexchange.json 2> $OUT/sharded_noexchange.err
EVS_DIRECT_A2A_V=0 timeout 300 python3 bench.py --gpus 1 --self-launch --force-sharded --force-exchange --steps 2000 --warmup 100 > $OUT/sharded_exchange_direct_sendrecv.json 2> /dev/null
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/sharded_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "ms_per_step %.4f"%j["ms_per_step"], j["config"]["exchange_mode"], j["config"].get("direct_a2a"), (j.get("single_process") or {}).get("sharded_over_single"), (j.get("single_process") or {}).get("ms_per_step"))
    except Exception as e: print(f, "ERR", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --gpus 1 --force-sharded --force-exchange --steps 2000 --warmup 100 > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-260 > $OUT/sharded_direct_kernel_stats.csv
rm -rf $OUT/trace
cat $OUT/sharded_direct_kernel_stats.csv
