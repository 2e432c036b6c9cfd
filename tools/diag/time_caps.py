"""Scratch diagnostic (GPU box): leap_cube plan-step time against the Newton iteration cap (fixed cost vs per-iteration cost)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from judo_amd import engine_model
from judo_amd.controller import make_controller
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for cap in [int(a) for a in sys.argv[2:]] or [0, 1, 2, 4, 8, 20]:
    engine_model.SOLVER_MAX_ITER = cap
    ctrl = make_controller('leap_cube', 'mppi'); ctrl.optimizer.config.num_rollouts = N; ctrl.controller_cfg.horizon = 0.64
    ctrl.reset(); ctrl.current_state = ctrl.task.default_state(); ctrl.system_metadata = {'goal_quat': np.array([0., 1, 0, 0])}
    for i in range(2): ctrl.update_action()
    torch.cuda.synchronize(); ctrl.model.stats(); t = time.perf_counter()
    for i in range(3): ctrl.time = 0.05 * i; ctrl.update_action()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3 * 1e3
    st = ctrl.model.stats()
    print(f'cap {cap:3d}  ms/plan {dt:7.2f}  iters/step {st["newton_iters"] / max(st["steps"], 1):.2f}', flush=True)
