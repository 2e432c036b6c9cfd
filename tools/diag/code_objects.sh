#!/bin/bash
# Register / scratch / LDS / occupancy of every kernel the product library ships (compiler remarks of the shipped flags; build container, no GPU).
# usage: tools/diag/code_objects.sh > profiles/rNN_code_objects.txt
cd "$(dirname "$0")/../.."
echo "# hipcc -Rpass-analysis=kernel-resource-usage on the shipped translation units with their .flags ($(git rev-parse --short HEAD), $(/opt/rocm/bin/hipcc --version | grep -m1 -o 'HIP version.*'))"
for f in jh_engine_v5 jh_engine_v5_cap64 jh_engine_v6 jh_engine_v4 jh_policy jh_simple jh_update jh_reward; do
  echo "## judo_amd/csrc/$f.hip  $(cat judo_amd/csrc/$f.flags 2>/dev/null)"
  tools/diag/resources.sh judo_amd/csrc/$f.hip 2>&1 | sed 's/^void *//' | grep -v "^Name:\s*$" | sed 's/ Name:$//'
done
