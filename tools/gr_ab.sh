# developer A/B of apply_emb alone: base vs variant libraries under ev-store-dlrm_amd/lib/var (tools/variants.sh)
cd $GRAFT_REPO_ROOT
for v in base "$@"; do
  if [ "$v" != base ]; then export EVS_LIB_PATH=$GRAFT_REPO_ROOT/ev-store-dlrm_amd/lib/var/libevstore_hip_$v.so; else unset EVS_LIB_PATH; fi
  echo "== $v"
  python3 tools/gather_bench.py 32 8 2>/dev/null | grep "^u"
done
