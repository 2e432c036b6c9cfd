#!/usr/bin/env python3
"""Golden vectors for the Spot task layer either side of the policy rollout (runs ONLY in the build container, needs /root/reference).

Recorded from the reference's own numpy code, imported through tools/_ref_import.py with a namespace object standing in for `self`
(the methods below touch only the attributes set here; constructing the real task needs MuJoCo, which this image lacks):

  S1  SpotBase.set_command_values / actuator_ctrlrange / nu      judo/tasks/spot/spot_base.py:166-263   (all 12 feature combinations)
  S2  SpotBase.apply_selection_mask + task_to_sim_ctrl           judo/tasks/spot/spot_base.py:265-391   (1-D, 2-D and 3-D controls)
  S3  SpotNavigate.reward                                        judo/tasks/spot/spot_navigate.py:50-77
  S4  override-resolved optimizer / controller configs           judo/optimizers/overrides.py:188-214, judo/controller/overrides.py:70-88
  S5  spot constants                                             judo/tasks/spot/spot_constants.py
"""

from __future__ import annotations

import itertools
import json
import os
import sys
from types import SimpleNamespace

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _ref_import  # noqa: E402

_ref_import.install()

from judo.controller.controller import ControllerConfig  # noqa: E402
from judo.optimizers.cem import CrossEntropyMethodConfig  # noqa: E402
from judo.optimizers.mppi import MPPIConfig  # noqa: E402
from judo.optimizers.ps import PredictiveSamplingConfig  # noqa: E402
from judo.tasks.spot import spot_constants as SC  # noqa: E402
from judo.tasks.spot.spot_base import SpotBase  # noqa: E402
from judo.tasks.spot.spot_navigate import SpotNavigate, SpotNavigateConfig  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _self(use_arm, use_gripper, use_legs, use_torso):
    s = SimpleNamespace(use_arm=use_arm, use_gripper=use_gripper, use_legs=use_legs, use_torso=use_torso, leg_selection_index=None, gripper_selection_index=None)
    SpotBase.set_command_values(s)
    s.default_policy_command = np.array([0, 0, 0] + list(SC.ARM_STOWED_POS) + [0] * 12 + [0, 0, SC.STANDING_HEIGHT_CMD])
    s.apply_selection_mask = lambda c: SpotBase.apply_selection_mask(s, c)
    return s


def main() -> None:
    out: dict[str, np.ndarray] = {}
    rng = np.random.default_rng(41)
    combos = [c for c in itertools.product((False, True), repeat=4) if not (c[1] and not c[0])]  # a gripper needs the arm
    out["combos"] = np.array(combos, dtype=np.int64)
    for ci, c in enumerate(combos):
        s = _self(*c)
        rngc = SpotBase.actuator_ctrlrange.fget(s)
        nu = len(s.default_command)
        out[f"c{ci}_default_command"] = s.default_command
        out[f"c{ci}_command_mask"] = s.command_mask
        out[f"c{ci}_ctrlrange"] = rngc
        ctl = rng.uniform(-1, 1, (6, 5, nu))
        if s.leg_selection_index is not None:  # exercise the three leg-selection bands and the exact thresholds
            ctl[:, :, s.leg_selection_index] = np.array([-0.9, -0.5, 0.0, 0.5, 0.51, 0.9])[:, None]
        if s.gripper_selection_index is not None:
            ctl[:, :, s.gripper_selection_index] = np.array([-0.3, 0.0, 0.3, -1.0, 1.0, 0.0])[:, None]
        out[f"c{ci}_controls"] = ctl
        out[f"c{ci}_sim3"] = SpotBase.task_to_sim_ctrl(s, ctl)
        out[f"c{ci}_sim2"] = SpotBase.task_to_sim_ctrl(s, ctl[:, 0])
        out[f"c{ci}_sim1"] = SpotBase.task_to_sim_ctrl(s, ctl[0, 0])
    # S3
    cfg = SpotNavigateConfig()
    cfg.goal_position = np.array([1.5, -0.5, SC.STANDING_HEIGHT])
    s = SimpleNamespace(config=cfg, model=SimpleNamespace(nq=26), body_pose_idx=0)
    states = rng.standard_normal((7, 9, 51)) * 0.5
    states[:, :, 2] = 0.5 + rng.standard_normal((7, 9)) * 0.1
    states[2, 4, 2] = 0.35   # exactly at the fallen threshold (<=)
    states[3, :, 2] = 0.6
    controls = rng.standard_normal((7, 9, 3))
    out["nav_states"], out["nav_controls"], out["nav_goal"] = states, controls, cfg.goal_position
    out["nav_reward"] = SpotNavigate.reward(s, states, None, controls)
    cfg.w_controls = 0.25
    out["nav_reward_wc"] = SpotNavigate.reward(s, states, None, controls)
    np.savez_compressed(os.path.join(OUT, "spot_tasks.npz"), **out)
    # S4 / S5
    res: dict = {"optimizer": {}, "controller": {}, "constants": {}}
    for task in ("spot_base", "spot_navigate"):
        res["optimizer"][task] = {}
        for name, cfg_cls in (("mppi", MPPIConfig), ("cem", CrossEntropyMethodConfig), ("ps", PredictiveSamplingConfig)):
            c = cfg_cls()
            c.set_override(task)
            res["optimizer"][task][name] = dict(vars(c))
        c = ControllerConfig()
        c.set_override(task)
        res["controller"][task] = dict(vars(c))
    res["task_defaults"] = {"spot_navigate": vars(SpotNavigateConfig())}
    for k in ("DEFAULT_SPOT_ROLLOUT_CUTOFF_TIME", "POLICY_OUTPUT_DIM", "LEGS_STANDING_POS", "LEGS_STANDING_POS_RL", "ARM_STOWED_POS", "ARM_UNSTOWED_POS", "STANDING_HEIGHT",
              "STANDING_HEIGHT_CMD", "BASE_SOFT_LIMITS", "TORSO_LOWER", "TORSO_UPPER", "GRIPPER_CLOSED_POS", "GRIPPER_OPEN_POS", "LEG_SOFT_LOWER_JOINT_LIMITS",
              "LEG_SOFT_UPPER_JOINT_LIMITS", "ARM_SOFT_LOWER_JOINT_LIMITS", "ARM_SOFT_UPPER_JOINT_LIMITS"):
        res["constants"][k] = getattr(SC, k)
    with open(os.path.join(OUT, "spot_configs.json"), "w") as f:
        json.dump(res, f, indent=1, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o))
    for fn in ("spot_tasks.npz", "spot_configs.json"):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
