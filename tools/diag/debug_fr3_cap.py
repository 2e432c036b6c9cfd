"""Scratch diagnostic (GPU box, -DJH_V3_EXITSTATS build): find a rollout of the first fr3_pick plan step whose constraint solve runs into the Newton iteration cap,
re-run that rollout alone, and hand the state at that step to the oracle (saved to gpurun_out/r3/cap_state.npz for study on the CPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np, torch
from judo_amd import _lib
from judo_amd.controller import make_controller
from judo_amd.rollout_backend import GpuRolloutBackend
from judo_amd.spline import evaluate
from oracle import oracle as O
c = make_controller("fr3_pick", "cem"); c.optimizer.config.num_rollouts = 32768; c.controller_cfg.horizon = 40 * c.task.dt
c.reset(); c.current_state = c.task.default_state(); c.optimizer.seed(1234); c.keep_candidates = True
x0 = c.current_state.copy(); t0 = c.time
c.update_action(); torch.cuda.synchronize()
L = _lib.lib(); L.jh_model_hist.argtypes = [C.c_void_p, C.POINTER(C.c_int)]; hh = (C.c_int * 40)(); L.jh_model_hist(c.model.handle, hh)
print("exits", list(hh)[:5], "first cap hit: flag", hh[15], "rollout", hh[16], "step", hh[17], c.model.stats())
n, step = hh[16], hh[17]
knots = c.candidate_knots_device[:, :, n].cpu().numpy().astype(np.float64)  # (K, nu)
H = c.num_timesteps
U = evaluate(c.spline_order, c.times, knots[None], c.times[0] + c.task.dt * np.arange(H))  # (1, H, nu)
be = GpuRolloutBackend("fr3_pick", 1)
gs, _, _ = be.rollout(x0, U)
print("alone:", be.model.stats())
om = O.Model("fr3_pick")
rs, _ = om.rollout(x0, U)
print("oracle vs kernel state error by step:", np.round(np.abs(gs - rs).max(axis=2)[0], 6).tolist())
xs = x0 if step == 0 else rs[0, step - 1]
o = om.forward(xs[:16], xs[16:], U[0, step])
print("oracle at the step: ncon", o["ncon"], "nefc", o["nefc"], "iters", o["solver_iter"], "fingers", xs[14:16], xs[29:31])
os.makedirs("gpurun_out/r3", exist_ok=True)
np.savez("gpurun_out/r3/cap_state.npz", x=xs, u=U[0, step], U=U, x0=x0, step=step, gs=gs, rs=rs)
