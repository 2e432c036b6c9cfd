"""Scratch diagnostic (GPU box): share of each phase of the cooperative leap_cube kernel, measured by repeating one phase
R times in a -DJH_V2_ABLATE build (tools/build_variant.sh ablate -DJH_V2_ABLATE) and differencing the plan-step times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from judo_amd import _lib, engine_model
from judo_amd.controller import make_controller
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
R = int(sys.argv[2]) if len(sys.argv) > 2 else 2
names = {0: 'baseline', 1: 'newton: assembly', 2: 'newton: assembly+factor', 3: 'newton: line search', 4: 'collision', 5: 'kinematics+dynamics'}
res = {}
for phase in [0, 1, 2, 3, 4, 5]:
    engine_model.ABLATE = (phase, R if phase else 1)
    ctrl = make_controller('leap_cube', 'mppi'); ctrl.optimizer.config.num_rollouts = N; ctrl.controller_cfg.horizon = 0.64
    ctrl.reset(); ctrl.current_state = ctrl.task.default_state(); ctrl.system_metadata = {'goal_quat': np.array([0., 1, 0, 0])}
    for i in range(2): ctrl.update_action()
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(3): ctrl.time = 0.05 * i; ctrl.update_action()
    torch.cuda.synchronize(); res[phase] = (time.perf_counter() - t) / 3 * 1e3
    extra = (res[phase] - res[0]) / max(R - 1, 1)
    print(f'{names[phase]:28s} {res[phase]:8.2f} ms' + (f'   phase cost {extra:7.2f} ms = {100 * extra / res[0]:5.1f}% of the plan step' if phase else ''), flush=True)
