#!/bin/bash
# On the GPU box, round 6 third pass: the resident dispatcher (tests + latency), the overlapped sharded step, the world-1 RCCL child test
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06c
mkdir -p $OUT
cd $ROOT
timeout 900 python3 -m pytest tests/test_gpu_serve.py -x -q -m gpu > $OUT/serve_test.log 2>&1; echo "serve test rc $?"; tail -15 $OUT/serve_test.log | cut -c1-400
timeout 600 python3 tools/serve_bench.py 1 128 2048 16384 > $OUT/serve_bench.log 2>&1; echo "serve bench rc $?"; tail -6 $OUT/serve_bench.log | cut -c1-300
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rccl_at_world_1 or extreme_codes" -s > $OUT/nccl_child.log 2>&1; echo "nccl child rc $?"; grep -E "passed|failed|worst|Error|assert" $OUT/nccl_child.log | tail -8 | cut -c1-300
for ov in "" "--overlap events" "--overlap signals"; do
  timeout 300 python3 bench.py --gpus 1 --self-launch --force-sharded --force-exchange --steps 2000 --warmup 100 $ov > $OUT/sharded_exchange_direct$ov.json 2> $OUT/err.txt; echo "direct $ov rc $?"; tail -c 300 $OUT/err.txt
  timeout 300 python3 bench.py --gpus 1 --self-launch --force-sharded --steps 2000 --warmup 100 $ov > $OUT/sharded_noexchange$ov.json 2> $OUT/err.txt; echo "noexchange $ov rc $?"
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/sharded_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "ms_per_step %.4f"%j["ms_per_step"], j["config"]["exchange_mode"], j["config"].get("direct_a2a"), "overlap", j["config"].get("overlap"), (j.get("single_process") or {}).get("sharded_over_single"), (j.get("single_process") or {}).get("ms_per_step"))
    except Exception as e: print(f, "ERR", e)
PY
