#!/bin/bash
# GPU box: A/B of library variants (build/libjudo_amd_<name>.so) on the recorded leap_cube plan inputs, hand self-collision on and off; with WRITE=1 also the HBM write traffic.
# usage: tools/gpu/ab_leap.sh name1 name2 ...
cd $GRAFT_REPO_ROOT
R="python tools/diag/ab_fixed_inputs.py replay tools/diag/ab_inputs_leap.npz"
for v in "$@"; do
  for self in 1 0; do
    [ "$SELFONLY" == "1" ] && [ $self == 0 ] && continue
    echo "== $v self=$self: $(JUDO_AMD_LIB=$PWD/build/libjudo_amd_$v.so SELF=$self $R 2>&1 | tail -2 | tr '\n' ' ' | sed 's/.*contacts dropped/dropped/; s/returned nominal.*//')"
  done
done
if [ "$WRITE" == "1" ]; then
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  for self in "" "--no-self-collision"; do
    rm -rf /tmp/pm; JUDO_AMD_LIB=$GRAFT_REPO_ROOT/build/libjudo_amd_$v.so rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pm -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-cube-only $self > /dev/null 2>&1
    echo "== $v $self WRITE_SIZE: $(python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/pm -name '*.db') | grep WRITE_SIZE | grep k_leap | awk '{print $(NF-3) " KB per launch"}')"
  done
done
fi
