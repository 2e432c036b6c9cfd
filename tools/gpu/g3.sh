cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fr3.py tests/test_gpu_controller.py tests/test_gpu_edges.py -x -q 2>&1 | tail -4
R="python tools/diag/ab_fixed_inputs.py replay tools/diag/ab_inputs_fr3.npz fr3_pick"
echo "== cur: $($R 2>&1 | tail -2 | tr '\n' ' ' | sed 's/.*contacts dropped/dropped/')"
cd /tmp; export TMPDIR=/tmp
for c in WRITE_SIZE FETCH_SIZE; do rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc $c -d /tmp/pm -o x -- python $GRAFT_REPO_ROOT/bench.py --task fr3_pick --steps 3 --warmup 2 --no-cpu-baseline --no-with-traces > /dev/null 2>&1
echo "== $c: $(python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/pm -name '*.db') | grep $c | grep k_fr3 | awk '{print $(NF-3) " KB per launch"}')"; done
