"""Evaluation sequences (alpha, slope, curvature) of line searches that hit the evaluation cap (needs a -DJH_V2_LSTRACE -DJH_V2_ITERDUMP build)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from judo_amd.controller import make_controller
N = 4096
c = make_controller("leap_cube", "mppi"); c.optimizer.config.num_rollouts = N; c.controller_cfg.horizon = 0.64
c.reset(); c.current_state = c.task.default_state(); c.system_metadata = {"goal_quat": np.array([0.0, 1.0, 0.0, 0.0])}; c.optimizer.seed(1234)
c.keep_candidates = True
c.update_action()
buf = c.candidate_knots_device.reshape(-1).cpu().numpy()
np.set_printoptions(linewidth=200, precision=4, suppress=False)
for s in range(12):
    o = buf[s * 40:(s + 1) * 40]
    print(f"case {s}: gp {o[0]:.3e} pMp {o[1]:.3e} pMd {o[2]:.3e} newton it {int(o[3])}")
    print("   alpha ", o[4::3][:12]); print("   d1/|gp|", o[5::3][:12] / abs(o[0])); print("   d2    ", o[6::3][:12])
