#!/bin/bash
# GPU box: the round-3 exhibits that are not bench lines -- VALU issue-rate microbenchmark, the occupancy experiment on the leap kernel (variants built by
# tools/diag/build_variant.sh), the 48-contact pool A/B.
cd $GRAFT_REPO_ROOT; out=gpurun_out/extras; mkdir -p $out
build/valu_rate > $out/valu_rate.txt 2>&1
{ echo "# cube contacts only, 16-contact pool, 4 knots (-DJH_V5_X_DIET -DJH_V5_NSLOT=1 -DJH_V5_NSBIG=1 -DJH_V5_MAXK=4 -DJH_V5_KNOTS_LDS=1 -DJH_V5_RSPAD=0); e4N: -DJH_V5_X_DYNRS, 256-thread workgroups (two waves per SIMD at run time), register budget of N waves per SIMD;"
  echo "# d123: 768-thread workgroup, three waves per SIMD resident"
  bash tools/gpu/ab_cube_only.sh e42 e43 e44 d123; } > $out/occupancy_experiment.txt 2>&1
{ echo "# base = round-3 kernel before the change (32-contact pool, knots in LDS); n3 = three slots per lane for every wave; ns3 = product (48-contact pool, second solver copy for waves above 32 contacts)"
  bash tools/gpu/ab_leap.sh base n3 ns3; } > $out/pool48_ab.txt 2>&1
cat $out/valu_rate.txt $out/occupancy_experiment.txt $out/pool48_ab.txt
