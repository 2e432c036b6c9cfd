#!/bin/bash
# Register / scratch / LDS / occupancy of every kernel in one translation unit (compiler remarks; no GPU needed).
# usage: tools/diag/resources.sh judo_amd/csrc/jh_engine_v3.hip [extra -D flags]
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Iinclude -Ijudo_amd/csrc $(cat "${f%.hip}.flags" 2>/dev/null) -c "$f" -o /tmp/res_$$.o \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|Spill|ScratchSize|Occupancy|LDS Size|SGPRs:" \
  | sed -e 's/.*remark: [^ ]* *//' -e 's/ \[-Rpass-analysis=kernel-resource-usage\]//' -e 's/Function Name: /\n/' | tr '\n' ' ' | sed 's/ _Z/\n_Z/g' | c++filt | sed 's/(.*)//'
echo
rm -f /tmp/res_$$.o
