# developer A/B by library on one box: the two- / three-tier chain (tools/c2bench.py), base = $1 (a library path), new = the tree's
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export EVS_LIB_PATH=$GRAFT_REPO_ROOT/$1; else unset EVS_LIB_PATH; fi
    echo "$v: $(python3 tools/c2bench.py 2>/dev/null | grep 'mixed-codec' | sed 's/.*consumer): //; s/ per batch.*//' | tr '\n' ' ')"
  done
done
