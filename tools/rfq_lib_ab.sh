# developer A/B by library on one box: the fused launch over reduced-precision tables (tools/kbench.py --bits), base = the
# tree's library, variants = names under lib/var (tools/variants.sh)
cd $GRAFT_REPO_ROOT
BITS=${BITS:-8}
for rep in 1 2; do
  for v in base "$@"; do
    if [ $v = base ]; then unset EVS_LIB_PATH; else export EVS_LIB_PATH=$GRAFT_REPO_ROOT/ev-store-dlrm_amd/lib/var/libevstore_hip_$v.so; fi
    python3 tools/kbench.py --bits $BITS --batch 16384 65536 --n-batches 16 2>/dev/null | grep "fused" | sed "s/^/$v: /" | cut -c1-170
  done
done
