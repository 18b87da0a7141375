# developer A/B by library on one box: the exact batch-1 GPU engine (tools/b1bench.py), base = tools/_build/libevstore_r04g.so
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in base new; do
  if [ $v = base ]; then export EVS_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/libevstore_r04g.so; else unset EVS_LIB_PATH; fi
  echo "$v: $(python3 tools/b1bench.py 2>/dev/null | tail -2 | tr '\n' ' ')"
done; done
