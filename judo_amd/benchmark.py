"""`python -m judo_amd.benchmark`: plan-step timing of every registered task x optimizer pair.

Mirror of the reference's `benchmark` entry point (judo/app/benchmark.py:16-107: for each pair collect `num_samples` plan times
and print mean +- std, median (IQR), min / max per task), run in-process on the GPU controller instead of through the dora graph.
Each pair uses the shipped per-task overrides (judo_amd/config.py), i.e. the rollout counts a judo user gets by default;
`--rollouts` overrides them for all pairs.
"""

from __future__ import annotations

import argparse
import json
import time

import numpy as np


def plan_times(task: str, optimizer: str, num_samples: int, warmup: int, rollouts: int | None) -> np.ndarray:
    import torch

    from judo_amd.controller import make_controller

    ctrl = make_controller(task, optimizer)
    if rollouts:
        ctrl.optimizer.config.num_rollouts = rollouts
    ctrl.reset()
    ctrl.current_state = ctrl.task.default_state()
    ctrl.system_metadata = ctrl.task.get_sim_metadata()
    out, t_plan = [], 0.0
    for i in range(warmup + num_samples):
        ctrl.time = t_plan
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctrl.update_action()  # ends with the device -> host copy of the new nominal: the plan is usable when it returns
        _ = ctrl.traces       # the reference's update_action ends with update_traces (controller.py:299): part of its plan time, so part of this one
        dt = time.perf_counter() - t0
        if i >= warmup:
            out.append(dt)
        t_plan += 1.0 / ctrl.controller_cfg.control_freq
    return np.array(out)


def summarize(t: np.ndarray) -> dict:
    return {"mean": float(t.mean()), "std": float(t.std()), "median": float(np.median(t)), "iqr25": float(np.percentile(t, 25)),
            "iqr75": float(np.percentile(t, 75)), "min": float(t.min()), "max": float(t.max())}


def main(argv: list[str] | None = None) -> dict:
    from judo_amd.optimizers import get_registered_optimizers
    from judo_amd.tasks import get_registered_tasks

    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("--num-samples", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rollouts", type=int, default=None)
    ap.add_argument("--tasks", nargs="*", default=None)
    ap.add_argument("--optimizers", nargs="*", default=None)
    ap.add_argument("--json", action="store_true", help="print one JSON object instead of the tables")
    a = ap.parse_args(argv)
    tasks = a.tasks or list(get_registered_tasks())
    opts = a.optimizers or list(get_registered_optimizers())
    results: dict[str, dict[str, dict]] = {}
    for task in tasks:
        for opt in opts:
            results.setdefault(task, {})[opt] = summarize(plan_times(task, opt, a.num_samples, a.warmup, a.rollouts))
    if a.json:
        print(json.dumps(results))
        return results
    for task, per_opt in results.items():
        print(f"\nResults for Task: {task}   (plan time, seconds)")
        print(f"  {'Optimizer':10s} {'Mean +- Std':>22s} {'Median (IQR)':>32s} {'Min / Max':>22s}")
        for opt, r in per_opt.items():
            print(f"  {opt:10s} {r['mean']:10.4f} +- {r['std']:.4f} {r['median']:12.4f} ({r['iqr25']:.4f}, {r['iqr75']:.4f}) {r['min']:12.4f} / {r['max']:.4f}")
    return results


if __name__ == "__main__":
    main()
