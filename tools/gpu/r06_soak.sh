#!/bin/bash
# GPU box: soak of the final build (600 closed-loop plan steps per configuration, every nominal and trace finite, solver counters per 100 steps) + 300 Spot plan steps
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6s; rm -rf $out; mkdir -p $out
timeout 900 python -W error::RuntimeWarning tools/diag/soak.py 2>&1 | grep -v amdgpu.ids | tee $out/soak.txt | tail -n 32
timeout 300 python -W error::RuntimeWarning - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a $out/soak.txt
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from judo_amd.controller import make_controller
for task in ("spot_navigate", "spot_base"):
    c = make_controller(task, "mppi"); c.reset(); c.current_state = c.task.default_state(); c.system_metadata = c.task.get_sim_metadata() if hasattr(c.task, "get_sim_metadata") else {}
    c.optimizer.seed(3); t = 0.0; t0 = time.perf_counter()
    for i in range(300):
        c.time = t; c.update_action(); tr = c.traces; t += 0.05
        assert np.isfinite(c.nominal_knots).all() and (tr is None or np.isfinite(tr).all()), (task, i)
    torch.cuda.synchronize()
    print(f"{task} mppi N={c.optimizer.num_rollouts}: 300 plan steps, {(time.perf_counter() - t0) / 300 * 1e3:.2f} ms/step, nominal finite; tree stats {c.policy_backend.engine.stats() if hasattr(c, 'policy_backend') and hasattr(c.policy_backend, 'engine') else ''}")
PY
