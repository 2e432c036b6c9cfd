"""Scratch diagnostic (GPU box): per-phase shader-cycle totals of the leap_cube engine, from an instrumented build
(build/libjudo_amd_prof.so, -DJH_ENGINE_PROFILE)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from judo_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), '..', 'build', 'libjudo_amd_prof.so')
from judo_amd.controller import make_controller
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctrl = make_controller('leap_cube', 'mppi')
ctrl.optimizer.config.num_rollouts = N
ctrl.controller_cfg.horizon = 0.64
ctrl.reset(); ctrl.current_state = ctrl.task.default_state()
ctrl.system_metadata = {'goal_quat': np.array([0., 1, 0, 0])}
for i in range(2):
    ctrl.update_action()
torch.cuda.synchronize(); ctrl.model.stats()
t = time.perf_counter(); ctrl.update_action(); torch.cuda.synchronize(); dt = time.perf_counter() - t
L = _lib.lib(); L.jh_model_profile.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
out = (C.c_longlong * 10)(); L.jh_model_profile(ctrl.model.handle, out)
names = ['kinematics', 'dynamics', 'collision', 'rows+warmstart', 'newton:assemble', 'newton:factor', 'newton:linesearch', 'integrate+cost']
tot = sum(out)
print(f'N={N} plan step {dt*1e3:.1f} ms; waves={N//4}; stats={ctrl.model.stats(reset=False)}')
for n, v in zip(names, out):
    print(f'  {n:12s} {v/ (N//4) / 64 / 1e3:10.1f} kcyc/step/wave  {100*v/max(tot,1):5.1f}%')

L.jh_model_hist.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
hh = (C.c_int * 40)(); L.jh_model_hist(ctrl.model.handle, hh)
nhh = list(hh)[24:40]; hh = list(hh)[:24]
tot = sum(hh)
print('broad-phase survivors per rollout-step:', ' '.join(f'{i}:{100*v/max(sum(nhh),1):.1f}%' for i, v in enumerate(nhh) if v))
print('newton iterations per solve (only steps with constraint rows):', ' '.join(f'{i}:{100*v/max(tot,1):.1f}%' for i, v in enumerate(hh) if v))
