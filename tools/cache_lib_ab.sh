# developer A/B by library on one box: the cache tier (300 unseen batches at capacity, events), base = the tree's library,
# variants = names under lib/var (tools/variants.sh)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in base "$@"; do
    if [ $v = base ]; then unset EVS_LIB_PATH; else export EVS_LIB_PATH=$GRAFT_REPO_ROOT/ev-store-dlrm_amd/lib/var/libevstore_hip_$v.so; fi
    python3 tools/cache_bench.py 16384 300 0 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v: %.2f us per batch (events %.2f) hit rate %.4f' % (r['ms_per_step']*1e3, r['roofline']['avg_launch_ms']*1e3, r['hit_rate']))"
  done
done
