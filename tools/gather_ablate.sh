#!/bin/bash
# On the GPU box: the gather-only launch (apply_emb alone, rows out as fp32) with its loads / its stores taken out
# (library variants -DEVS_GR_NOLOAD / -DEVS_GR_NOSTORE, tools/variants.sh ...@evs_gather), u8 and fp32 tables.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in base noload nostore; do
  if [ $v = base ]; then unset EVS_LIB_PATH; else export EVS_LIB_PATH=$R/ev-store-dlrm_amd/lib/var/libevstore_hip_$v.so; fi
  echo "== $v"
  python3 $R/tools/kbench.py --bits 8 --codes encoded --batch ${B:-16384 65536} 2>/dev/null | grep fused | sed 's/.*| gather only/u8 gather only/'
  python3 $R/tools/kbench.py --batch ${B:-16384 65536} 2>/dev/null | grep "gather tile" | sed 's/|.*//'
done
