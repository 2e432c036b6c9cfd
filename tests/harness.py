"""Test-side restatement of one plan step (`Controller.update_action`, judo/controller/controller.py:246-293) built
ONLY from oracle primitives (oracle/oracle.py): per-knot sigma -> sample (injected noise) -> clip -> spline ->
rollout (fp64 engine) -> reward -> update.  Used by the GPU parity tests and by __graft_entry__.smoke()."""

from __future__ import annotations

import numpy as np

from oracle import oracle as O


def oracle_reward(task, states, sensors, controls, system_metadata=None) -> np.ndarray:
    name = task.name
    if name == "cartpole":
        c = task.config
        return O.reward_cartpole(states, controls, (c.w_vertical, c.w_centered, c.w_velocity, c.w_control, c.p_vertical, c.p_centered))
    if name == "cylinder_push":
        c = task.config
        return O.reward_cylinder(states, (c.w_pusher_proximity, c.w_pusher_velocity, c.w_cart_position, c.pusher_goal_offset, c.goal_pos[0], c.goal_pos[1]))
    if name == "leap_cube":
        gq = (system_metadata or {}).get("goal_quat", np.array([1.0, 0, 0, 0]))
        return O.reward_leap(states, gq, task.config.w_pos, task.config.w_rot, task.goal_pos)
    if name == "fr3_pick":
        p = task.task_params()
        return O.reward_fr3(states, sensors, task.phase, p[:13], p[13:22])
    raise KeyError(name)


def oracle_knot_sigma(opt_name: str, cfg, nu: int, cem_sigma: np.ndarray | None = None) -> np.ndarray:
    K = cfg.num_nodes
    if opt_name in ("mppi", "ps"):
        return O.mppi_sigma(cfg.sigma, cfg.use_noise_ramp, cfg.noise_ramp, K, nu)
    assert cem_sigma is not None
    return O.cem_sigma_ramp(cem_sigma, cfg.use_noise_ramp, cfg.noise_ramp, cfg.sigma_min, cfg.sigma_max)


def oracle_plan_step(om: "O.Model", ctrl, nominal_shifted: np.ndarray, noise: np.ndarray, opt_name: str | None = None,
                     cem_sigma: np.ndarray | None = None, nthread: int | None = None) -> dict:
    """One optimiser iteration around `nominal_shifted` (K,nu) with reference-layout noise (N-1,K,nu)."""
    task, cfg = ctrl.task, ctrl.optimizer.config
    opt_name = opt_name or {"GpuMPPI": "mppi", "GpuCEM": "cem", "GpuPS": "ps"}[type(ctrl.optimizer).__name__]
    K, nu, H = cfg.num_nodes, task.nu, ctrl.num_timesteps
    sigma = oracle_knot_sigma(opt_name, cfg, nu, cem_sigma)
    knots = O.sample_knots(nominal_shifted, np.asarray(noise, dtype=np.float64), sigma)
    r = task.actuator_ctrlrange
    knots = O.clip_knots(knots, r[:, 0], r[:, 1])
    W = O.spline_weights(ctrl.spline_order, ctrl.spline_timesteps, ctrl.rollout_times)
    U = O.spline_eval(W, knots)
    task.pre_rollout(ctrl.current_state)
    states, sensors = om.rollout(ctrl.current_state, U, nthread)
    rewards = oracle_reward(task, states, sensors, U, ctrl.system_metadata)
    out = dict(knots=knots, U=U, states=states, sensors=sensors, rewards=rewards, sigma_used=sigma)
    if opt_name == "mppi":
        out["nominal"] = O.mppi_update(knots, rewards, cfg.temperature)
    elif opt_name == "ps":
        out["nominal"] = O.ps_update(knots, rewards)
    else:
        out["nominal"], out["sigma"], out["elite_idx"] = O.cem_update(knots, rewards, cfg.num_elites, cfg.sigma_min, cfg.sigma_max)
    return out


def oracle_update_action(opt_name: str, cfg, ccfg, nu: int, dt: float, ctrlrange: np.ndarray, rollout_fn, reward_fn, state: dict, x0: np.ndarray, time: float,
                         noises: list[np.ndarray], trace_adrs=()) -> dict:
    """The whole of `Controller.update_action` (judo/controller/controller.py:210-299) from oracle primitives, for any rollout / reward pair.

    `state` carries what the controller object carries between plan steps: times, nominal_knots (raw), normalizer (O.OracleNormalizer), cem sigma.
    `noises[i]` is the (N-1, K, nu) standard-normal draw of iteration i.  Returns the plan step's outputs and updates `state` in place."""
    K = cfg.num_nodes
    H = int(np.ceil(ccfg.horizon / dt))
    order = ccfg.spline_order
    new_times = time + np.linspace(0, ccfg.horizon, K, endpoint=True)
    nrm = state["normalizer"]
    nominal_n = nrm.normalize(O.spline_resample(order, state["times"], state["nominal_knots"], new_times))
    if opt_name == "cem" and len(state["sigma"]) != K:
        state["sigma"] = O.cem_pre_optimization(state["sigma"], state["times"], new_times)
    W = O.spline_weights(order, new_times, time + dt * np.arange(H))
    out = {}
    for i in range(ccfg.max_opt_iters):
        if opt_name == "cem":
            state["sigma"] = O.cem_sigma_ramp(state["sigma"], cfg.use_noise_ramp, cfg.noise_ramp, cfg.sigma_min, cfg.sigma_max)  # cumulative (cem.py:69-72)
            sigma = state["sigma"]
        else:
            sigma = O.mppi_sigma(cfg.sigma, cfg.use_noise_ramp, cfg.noise_ramp, K, nu)
        cand_n = O.sample_knots(nominal_n, np.asarray(noises[i], dtype=np.float64), sigma)
        cand_n = O.clip_knots(cand_n, nrm.normalize(ctrlrange[:, 0]), nrm.normalize(ctrlrange[:, 1]))
        cand = nrm.denormalize(cand_n)
        U = O.spline_eval(W, cand)
        states, sensors = rollout_fn(x0, U)
        rewards = reward_fn(states, sensors, U)
        if opt_name == "mppi":
            nominal_n = O.mppi_update(cand_n, rewards, cfg.temperature)
        elif opt_name == "ps":
            nominal_n = O.ps_update(cand_n, rewards)
        else:
            nominal_n, state["sigma"], _ = O.cem_update(cand_n, rewards, cfg.num_elites, cfg.sigma_min, cfg.sigma_max)
        nrm.update(cand)
        out.update(candidates=cand, controls=U, states=states, sensors=sensors, rewards=rewards)
    state["nominal_knots"] = nrm.denormalize(nominal_n)
    state["times"] = new_times
    out["nominal"] = state["nominal_knots"].copy()
    if trace_adrs:
        out["traces"] = O.trace_segments(out["sensors"], out["rewards"], trace_adrs, ccfg.max_num_traces)
    return out
