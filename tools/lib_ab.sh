# developer A/B by library on one box: the headline bench line (events), base = the tree's library, variants = names under lib/var
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in base "$@"; do
    if [ $v = base ]; then unset EVS_LIB_PATH; else export EVS_LIB_PATH=$GRAFT_REPO_ROOT/ev-store-dlrm_amd/lib/var/libevstore_hip_$v.so; fi
    python3 bench.py --steps 2000 --warmup 500 --no-cpu-baseline --no-extras --no-cache-tier 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$v: %.2f us (events %.2f) declared %.2f' % (j['ms_per_step']*1e3, j['roofline']['avg_launch_ms']*1e3, j['declared_one_index']['ms_per_step']*1e3))"
  done
done
