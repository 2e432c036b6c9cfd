#!/usr/bin/env python3
"""Pin the physics oracle to MuJoCo itself -- runs only where the `mujoco` wheel (3.5.0, the reference's pin) and the reference's MJCF are
reachable; in this image neither is, so the fixtures it writes do not exist yet and tests/test_physics_golden.py skips.

    python tools/gen_golden_mujoco.py            # -> tests/golden/physics_<task>.npz for every task whose MJCF loads

Per task: `mj_step` trajectories from the reference's `MJRolloutBackend` call pattern (oracle/mujoco_probe.py) for seeded controls from the task's
reset pose and from perturbed states: x0, controls, states (N, H, nq+nv), sensordata (N, H, ns), plus one-step probes that isolate the three MuJoCo
facts the oracle could only take from the documentation: the sensor lag (sensordata[h] describes the state before step h), the
pyramidal friction regularisation with mu clamped to its minimum (cylinder_push: contact force along the line of centres), and the soft joint limit.
cartpole / cylinder_push need no meshes and are generated here; leap_cube / fr3_pick reference mesh assets that are not in the repository
(`.MISSING_LARGE_BLOBS`), so their fixtures can only come from a checkout that has them -- add the task to CASES there.
"""

from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mujoco_probe as MP  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
CASES = {
    # task: (x0, control scale, N, H)
    "cartpole": (np.array([1.0, np.pi, 0.0, 0.0]), 1.5, 16, 64),
    "cylinder_push": (np.array([0.0, 0.0, 0.45, 0.1, 0.5, 0, 0, 0]), 1.0, 16, 64),  # starts in contact
}


def main() -> int:
    if MP.find_mujoco() is None:
        print("mujoco is not installed: nothing generated (tests/test_physics_golden.py keeps skipping)")
        return 1
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(2024)
    wrote = 0
    for task, (x0, scale, N, H) in CASES.items():
        if MP.reference_xml(task) is None:
            print(f"{task}: MJCF not reachable (set JUDO_REFERENCE_ROOT), skipped")
            continue
        nu = {"cartpole": 1, "cylinder_push": 2}[task]
        U = np.repeat(rng.standard_normal((N, H // 4, nu)) * scale, 4, axis=1)
        states, sensors = MP.rollout(task, x0, U, nthread=1)
        xb = x0[None] + 0.05 * rng.standard_normal((N, x0.size))
        states_b, sensors_b = MP.rollout(task, xb, U, nthread=1)
        np.savez_compressed(os.path.join(OUT, f"physics_{task}.npz"), x0=x0, controls=U, states=states, sensors=sensors, x0_batched=xb,
                            states_batched=states_b, sensors_batched=sensors_b, mujoco_version=np.array(MP.find_mujoco().__version__))
        print(f"{task}: wrote tests/golden/physics_{task}.npz ({N} x {H} steps)")
        wrote += 1
    return 0 if wrote else 1


if __name__ == "__main__":
    raise SystemExit(main())
