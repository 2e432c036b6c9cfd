#!/bin/bash
# Assembly listing with line tables of the leap kernel's translation unit (build/isa/v5.s), then the hot-loop summaries of both solver copies.  usage: tools/diag/isa_v5.sh [extra -D flags]
cd "$(dirname "$0")/../.."
mkdir -p build/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Iinclude -Ijudo_amd/csrc $(cat judo_amd/csrc/jh_engine_v5.flags) -gline-tables-only -S --cuda-device-only "$@" judo_amd/csrc/jh_engine_v5.hip -o build/isa/v5.s 2>/dev/null
for w in lean hand; do python tools/diag/isa_hot_loop.py build/isa/v5.s $w | head -2; python tools/diag/isa_dump_loop.py build/isa/v5.s $w > build/isa/v5_$w.txt; echo "  branches $(grep -c 's_cbranch\|s_branch' build/isa/v5_$w.txt) saveexec $(grep -c saveexec build/isa/v5_$w.txt) scratch $(grep -c scratch_ build/isa/v5_$w.txt) readlane $(grep -c 'v_readlane\|v_writelane' build/isa/v5_$w.txt)"; done
grep -A12 "^_ZN.*k_leap_v5ILb0ELi4ELb1.*:$" build/isa/v5.s > /dev/null; grep -E "\.(sgpr|vgpr)_spill_count|scratch_en|\.private_segment_fixed_size" build/isa/v5.s | head -12
