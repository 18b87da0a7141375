#!/bin/bash
# Developer A/B: build libevstore_hip variants with extra -D flags for ONE source file (default evs_fused).
# usage: tools/variants.sh name1:"-DFLAG ..." name2@evs_cache:"-DFLAG"  ->  ev-store-dlrm_amd/lib/var/libevstore_hip_<name>.so
set -e
cd "$(dirname "$0")/../ev-store-dlrm_amd/csrc"
make -s
mkdir -p ../lib/var
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -I../../include"
for spec in "$@"; do
  nf="${spec%%:*}"; defs="${spec#*:}"
  name="${nf%%@*}"; file="evs_fused"; [[ "$nf" == *@* ]] && file="${nf#*@}"
  /opt/rocm/bin/hipcc $FLAGS $defs -c $file.hip -o ../lib/var/${file}_$name.o
  objs=$(ls ../lib/obj/*.o | grep -v "/$file.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/var/libevstore_hip_$name.so $objs ../lib/var/${file}_$name.o
  echo built $name
done
