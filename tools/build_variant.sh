#!/bin/bash
# usage: tools/build_variant.sh <name> [extra hipcc flags...]   -> build/libjudo_amd_<name>.so
# Builds a variant of the product library with extra -D flags (scratch experiments; the product build is __graft_entry__.build()).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/var_$name
pids=()
for s in jh_api jh_simple jh_update jh_engine jh_engine_v2 jh_engine_v5 jh_engine_v3 jh_engine_v4 jh_policy; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Iinclude -Ijudo_amd/csrc "$@" -c judo_amd/csrc/$s.hip -o build/var_$name/$s.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libjudo_amd_$name.so build/var_$name/*.o
echo build/libjudo_amd_$name.so
