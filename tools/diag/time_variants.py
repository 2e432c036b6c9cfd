"""Scratch diagnostic (GPU box): plan-step time of a library variant given by JH_LIB."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from judo_amd import _lib
if os.environ.get("JH_LIB"): _lib.LIB_PATH = os.environ["JH_LIB"]
from judo_amd.controller import make_controller
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ctrl = make_controller('leap_cube', 'mppi'); ctrl.optimizer.config.num_rollouts = N; ctrl.controller_cfg.horizon = 0.64
ctrl.reset(); ctrl.current_state = ctrl.task.default_state(); ctrl.system_metadata = {'goal_quat': np.array([0., 1, 0, 0])}
for i in range(2): ctrl.update_action()
torch.cuda.synchronize(); t = time.perf_counter()
for i in range(5): ctrl.time = 0.05 * i; ctrl.update_action()
torch.cuda.synchronize(); print(os.environ.get("JH_LIB"), 'N', N, 'ms/plan', (time.perf_counter() - t) / 5 * 1e3, ctrl.model.stats())
