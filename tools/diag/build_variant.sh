#!/bin/bash
# Build a variant of the PRODUCT library that differs from the default in ONE translation unit (other objects come from build/*.o of the last default build).
# usage: tools/diag/build_variant.sh <name> <file.hip> [extra compiler flags]   ->  variants/libjudo_amd_<name>.so   (select it with JUDO_AMD_LIB)
# VARIANT_FLAGS="..." replaces the translation unit's .flags line (backend options).
# variants/ is git-ignored but travels with the gpurun snapshot (build/ does not: .gpurunignore).
set -e
cd "$(dirname "$0")/../.."
name=$1; src=$2; shift 2
base=$(basename "$src" .hip)
mkdir -p variants build/var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Iinclude -Ijudo_amd/csrc ${VARIANT_FLAGS-$(cat "${src%.hip}.flags" 2>/dev/null)} -c "$src" -o "build/var/${base}_${name}.o" "$@"
objs=""
for o in jh_api jh_simple jh_update jh_reward jh_engine_v5 jh_engine_v5_cap64 jh_engine_v6 jh_engine_v4 jh_policy; do
  if [ "$o" == "$base" ]; then objs="$objs build/var/${base}_${name}.o"; else objs="$objs build/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "variants/libjudo_amd_${name}.so" $objs
echo "variants/libjudo_amd_${name}.so"
