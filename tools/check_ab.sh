# developer A/B by library on one box: the headline launch (lS_o given), the u8 fused launch and apply_emb alone
cd $GRAFT_REPO_ROOT
BASE=$GRAFT_REPO_ROOT/tools/_build/libevstore_r04b.so
for rep in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export EVS_LIB_PATH=$BASE; else unset EVS_LIB_PATH; fi
    h=$(python3 bench.py --steps 2000 --warmup 500 --no-cpu-baseline --no-extras --no-cache-tier 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('%.2f us (events %.2f)' % (j['ms_per_step']*1e3, j['roofline']['avg_launch_ms']*1e3))")
    q=$(python3 tools/kbench.py --fused-only --bits 8 --codes encoded --batch 16384 65536 2>/dev/null | grep offsets | sed 's/.*| offsets *//' | tr '\n' ' ')
    g=$(python3 tools/gather_bench.py 32 B=16384 2>/dev/null | grep "^u" | sed 's/=.*//' )
    echo "$v: headline $h | u8 offsets-given $q | $g"
  done
done
