#!/bin/bash
# Developer A/B: build libevstore_hip variants with extra -D flags for evs_fused.hip.
# usage: tools/variants.sh name1:"-DFLAG ..." name2:"..."   ->  ev-store-dlrm_amd/lib/var/libevstore_hip_<name>.so
set -e
cd "$(dirname "$0")/../ev-store-dlrm_amd/csrc"
make -s
mkdir -p ../lib/var
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -I../../include"
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"
  /opt/rocm/bin/hipcc $FLAGS $defs -c evs_fused.hip -o ../lib/var/fused_$name.o
  objs=$(ls ../lib/obj/*.o | grep -v evs_fused)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/var/libevstore_hip_$name.so $objs ../lib/var/fused_$name.o
  echo built $name
done
