"""Task plugin surface (mirror of judo/tasks/base.py:24-203 for the hot path) and the four BASELINE tasks.

A task contributes: the model constants (JSON description -> device blob), `dt`, `nu`, `actuator_ctrlrange`,
the cost weights as a small float vector (`task_params`) that the fused kernel reads from LDS, the host-side
hooks `pre_rollout` (fr3_pick phase decision) / `optimizer_warm_start` / `reset`, and `reward(...)` with the
reference's signature (evaluated on the device by `jh_task_reward`).

Reference cost definitions: judo/tasks/cartpole.py:42-78, cylinder_push.py:50-93, leap_cube.py:63-88,
fr3_pick.py:225-311; config defaults :20-28, :20-36, :30-35, :44-101.
"""

from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass, field
from enum import Enum
from typing import Any, Generic, TypeVar

import numpy as np

from judo_amd.models import actuator_ctrlrange, layout, load_description, qpos0


@dataclass
class TaskConfig:
    """Base task configuration dataclass."""


ConfigT = TypeVar("ConfigT", bound=TaskConfig)


@dataclass
class TaskData:
    """The slice of MjData the hot path touches (judo/tasks/base.py:38, :42-50)."""

    qpos: np.ndarray
    qvel: np.ndarray
    time: float = 0.0


class Task(ABC, Generic[ConfigT]):
    name: str
    config_t: type
    reward_accepts_torch: bool = True  # a plugin whose overridden `reward` is numpy-only sets this to False

    def __init__(self) -> None:
        self.desc = load_description(getattr(self, "model_name", None) or self.name)  # several tasks can share one model (the Spot tasks)
        self._layout = layout(self.desc)
        self.config = self.config_t()
        self.data = TaskData(qpos0(self.desc), np.zeros(self._layout.nv))
        self.phase = 0
        self._gpu = None  # GpuModel, created on first device use

    # ---- model facts -------------------------------------------------------------------------------------
    @property
    def nq(self) -> int:
        return self._layout.nq

    @property
    def nv(self) -> int:
        return self._layout.nv

    @property
    def nu(self) -> int:
        return self._layout.nu

    @property
    def nsensordata(self) -> int:
        return self._layout.ns

    @property
    def dt(self) -> float:
        return float(self.desc["option"]["timestep"]) * self.physics_substeps

    @property
    def physics_substeps(self) -> int:
        return 1

    @property
    def time(self) -> float:
        return self.data.time

    @time.setter
    def time(self, value: float) -> None:
        self.data.time = value

    @property
    def actuator_ctrlrange(self) -> np.ndarray:
        if getattr(self, "_ctrlrange", None) is None:
            self._ctrlrange = actuator_ctrlrange(self.desc)
            self._ctrlrange.setflags(write=False)
        return self._ctrlrange

    # ---- indices into the `states` / `sensors` arrays by name (judo/tasks/base.py:180-204: what a plugin's `reward` uses to find its columns) ----
    def _joint_addresses(self) -> dict[str, tuple[int, int]]:
        if getattr(self, "_jadr", None) is None:
            nq_of, nv_of = {"free": 7, "ball": 4, "slide": 1, "hinge": 1}, {"free": 6, "ball": 3, "slide": 1, "hinge": 1}
            out, q, v = {}, 0, 0
            for j in self.desc["joints"]:  # MuJoCo lays qpos / qvel out joint by joint in declaration (= body tree) order
                out[j["name"]] = (q, v)
                q += nq_of[j["type"]]
                v += nv_of[j["type"]]
            self._jadr = out
        return self._jadr

    def get_sensor_start_index(self, sensor_name: str) -> int:
        """First column of the named sensor in the `sensors` array (`model.sensor(name).adr[0]`)."""
        for sn in self.desc["sensors"]:
            if sn["name"] == sensor_name:
                return int(sn["adr"])
        raise KeyError(f"Invalid name '{sensor_name}'. Valid names: {[sn['name'] for sn in self.desc['sensors']]}")

    def get_joint_position_start_index(self, joint_name: str) -> int:
        """First column of the named joint's position in the `states` array (`model.jnt_qposadr`)."""
        try:
            return self._joint_addresses()[joint_name][0]
        except KeyError:
            raise KeyError(f"Invalid name '{joint_name}'. Valid names: {list(self._joint_addresses())}") from None

    def get_joint_velocity_start_index(self, joint_name: str) -> int:
        """First column of the named joint's velocity in the `states` array: AFTER the nq position columns (`nq + model.jnt_dofadr`)."""
        try:
            return self.nq + self._joint_addresses()[joint_name][1]
        except KeyError:
            raise KeyError(f"Invalid name '{joint_name}'. Valid names: {list(self._joint_addresses())}") from None

    @property
    def locomotion_policy_path(self) -> str | None:
        """judo/tasks/base.py:83-90: path of the ONNX locomotion policy of a policy task (the Spot tasks override it), None otherwise."""
        return None

    @property
    def uses_locomotion_policy(self) -> bool:
        """judo/tasks/base.py:92-95: a task is a policy task exactly when it names a policy."""
        return self.locomotion_policy_path is not None

    def pre_sim_step(self) -> None:
        """judo/tasks/base.py:140-144: hooks of the SIMULATION node around its own `mj_step`; the plan step never calls them (kept so that a plugin task written
        against the reference's base class subclasses this one unchanged)."""

    def post_sim_step(self) -> None:
        pass

    def gpu_model(self, device=None):
        from judo_amd.device import GpuModel

        if self._gpu is None or (device is not None and self._gpu.device != device):
            self._gpu = GpuModel(self.desc, device)
        return self._gpu

    # ---- hooks ---------------------------------------------------------------------------------------------
    def pre_rollout(self, curr_state: np.ndarray) -> None:
        """Pre-rollout behaviour (no-op by default)."""

    def post_rollout(self, states, sensors, controls, system_metadata=None) -> None:
        """Post-rollout behaviour (no-op by default)."""

    def optimizer_warm_start(self) -> np.ndarray:
        return np.zeros(self.nu)

    def task_to_sim_ctrl(self, controls):
        return controls

    def get_sim_metadata(self) -> dict[str, Any]:
        return {}

    def reset(self) -> None:
        self.data.qpos = np.zeros(self.nq)
        self.data.qvel = np.zeros(self.nv)

    def default_state(self) -> np.ndarray:
        """Deterministic x0 used by the benchmark / parity harness (SURVEY.md section 8d "synthetic inputs")."""
        return np.concatenate([self.data.qpos, self.data.qvel])

    # ---- cost ----------------------------------------------------------------------------------------------
    @abstractmethod
    def task_params(self, system_metadata: dict[str, Any] | None = None) -> np.ndarray:
        """Cost weights / goals as the fp32 vector the kernels stage in LDS."""

    def reward(self, states, sensors, controls, system_metadata: dict[str, Any] | None = None):
        """Task.reward(states (N,H,nx), sensors (N,H,ns), controls (N,H,nu)) -> (N,), evaluated on the GPU.

        numpy in -> numpy (float64) out; torch device tensors in -> torch device tensor out."""
        import torch

        from judo_amd import _lib
        from judo_amd.device import current_stream_ptr, f32

        gm = self.gpu_model()
        as_numpy = not isinstance(states, torch.Tensor)
        st = f32(states, gm.device) if as_numpy else states.to(torch.float32).contiguous()
        if st.ndim != 3 or st.shape[-1] != gm.nx:
            raise ValueError(f"states must be (N, H, {gm.nx}), got {tuple(st.shape)}")
        N, H = int(st.shape[0]), int(st.shape[1])
        se = None
        if sensors is not None:
            se = f32(sensors, gm.device) if not isinstance(sensors, torch.Tensor) else sensors.to(torch.float32).contiguous()
        co = None
        if controls is not None:
            co = f32(controls, gm.device) if not isinstance(controls, torch.Tensor) else controls.to(torch.float32).contiguous()
        tp = f32(self.task_params(system_metadata), gm.device)
        out = torch.empty(N, dtype=torch.float32, device=gm.device)
        s = _lib.lib().jh_task_reward(gm.handle, _lib.ptr(st), _lib.ptr(se), _lib.ptr(co), _lib.ptr(tp), int(self.phase), N, H, _lib.ptr(out), current_stream_ptr())
        _lib.check(s, "jh_task_reward")
        return out.cpu().numpy().astype(np.float64) if as_numpy else out


# ------------------------------------------------------------------------------------------------ cartpole
@dataclass
class CartpoleConfig(TaskConfig):
    w_vertical: float = 10.0
    w_centered: float = 10.0
    w_velocity: float = 0.1
    w_control: float = 0.1
    p_vertical: float = 0.01
    p_centered: float = 0.1


class Cartpole(Task[CartpoleConfig]):
    name = "cartpole"
    config_t = CartpoleConfig

    def __init__(self) -> None:
        super().__init__()
        self.reset()

    def task_params(self, system_metadata=None) -> np.ndarray:
        c = self.config
        return np.array([c.w_vertical, c.w_centered, c.w_velocity, c.w_control, c.p_vertical, c.p_centered], dtype=np.float32)

    def reset(self) -> None:  # judo/tasks/cartpole.py:80-84
        self.data.qpos = np.array([1.0, np.pi]) + np.random.randn(2)
        self.data.qvel = 1e-1 * np.random.randn(2)

    def default_state(self) -> np.ndarray:
        return np.array([1.0, np.pi, 0.0, 0.0])


# ------------------------------------------------------------------------------------------------ cylinder_push
@dataclass
class CylinderPushConfig(TaskConfig):
    w_pusher_proximity: float = 0.5
    w_pusher_velocity: float = 0.0
    w_cart_position: float = 0.1
    pusher_goal_offset: float = 0.25
    goal_pos: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0]))


class CylinderPush(Task[CylinderPushConfig]):
    name = "cylinder_push"
    config_t = CylinderPushConfig

    def __init__(self) -> None:
        super().__init__()
        self.reset()

    def task_params(self, system_metadata=None) -> np.ndarray:
        c = self.config
        return np.array([c.w_pusher_proximity, c.w_pusher_velocity, c.w_cart_position, c.pusher_goal_offset, c.goal_pos[0], c.goal_pos[1]], dtype=np.float32)

    def reset(self) -> None:  # judo/tasks/cylinder_push.py:95-107
        theta = 2 * np.pi * np.random.rand(2)
        self.data.qpos = np.array([np.cos(theta[0]), np.sin(theta[0]), 2 * np.cos(theta[1]), 2 * np.sin(theta[1])])
        self.data.qvel = np.zeros(4)

    def default_state(self) -> np.ndarray:
        return np.array([1.0, 0.0, 2 * np.cos(1.0), 2 * np.sin(1.0), 0, 0, 0, 0])


# ------------------------------------------------------------------------------------------------ leap_cube
LEAP_QPOS_HOME = np.array(
    [0.0, 0.03, 0.1, 1.0, 0.0, 0.0, 0.0, 0.5, -0.75, 0.75, 0.25, 0.5, 0.0, 0.75, 0.25, 0.5, 0.75, 0.75, 0.25, 0.65, 0.9, 0.75, 0.6]
)  # judo/tasks/leap_cube.py:16-24


@dataclass
class LeapCubeConfig(TaskConfig):
    w_pos: float = 100.0
    w_rot: float = 0.1


class LeapCube(Task[LeapCubeConfig]):
    name = "leap_cube"
    config_t = LeapCubeConfig

    def __init__(self) -> None:
        super().__init__()
        self.goal_pos = np.array([0.0, 0.03, 0.1])
        self.goal_quat = np.array([1.0, 0.0, 0.0, 0.0])
        self.qpos_home = LEAP_QPOS_HOME
        self.reset_command = LEAP_QPOS_HOME[7:].copy()
        self.reset()

    def task_params(self, system_metadata=None) -> np.ndarray:
        gq = (system_metadata or {}).get("goal_quat", np.array([1.0, 0.0, 0.0, 0.0]))  # leap_cube.py:73
        return np.array([self.config.w_pos, self.config.w_rot, *self.goal_pos, *gq], dtype=np.float32)

    def optimizer_warm_start(self) -> np.ndarray:
        return self.reset_command.copy()

    def _update_goal_quat(self) -> None:  # leap_cube.py:111-125 (uniform random unit quaternion)
        u = np.random.rand(3)
        self.goal_quat = np.array([np.sqrt(1 - u[0]) * np.sin(2 * np.pi * u[1]), np.sqrt(1 - u[0]) * np.cos(2 * np.pi * u[1]),
                                   np.sqrt(u[0]) * np.sin(2 * np.pi * u[2]), np.sqrt(u[0]) * np.cos(2 * np.pi * u[2])])

    def reset(self) -> None:
        self.data.qpos = self.qpos_home.copy()
        self.data.qvel = np.zeros(self.nv)
        self._update_goal_quat()

    def get_sim_metadata(self) -> dict[str, Any]:
        return {"goal_quat": self.goal_quat}

    def default_state(self) -> np.ndarray:
        return np.concatenate([LEAP_QPOS_HOME, np.zeros(22)])


LEAP_DOWN_QPOS_HOME = np.array(
    [-0.04, -0.035, -0.065, 1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.8, 0.8, 1.0, 0.0, 0.8, 0.8, 1.0, 0.0, 0.8, 0.8, 1.0, 1.0, 0.4, 0.9]
)  # judo/tasks/leap_cube_down.py:14-22


@dataclass
class LeapCubeDownConfig(LeapCubeConfig):
    w_rot: float = 0.05


class LeapCubeDown(LeapCube):
    """Palm-down variant (judo/tasks/leap_cube_down.py:33-53): same hand and cube, identity hand orientation, different home
    pose and goal position; runs on the leap_cube kernels with its own model constants."""

    name = "leap_cube_down"
    config_t = LeapCubeDownConfig

    def __init__(self) -> None:
        Task.__init__(self)
        self.goal_pos = np.array([-0.04, -0.035, -0.065])
        self.goal_quat = np.array([1.0, 0.0, 0.0, 0.0])
        self.qpos_home = LEAP_DOWN_QPOS_HOME
        self.reset_command = LEAP_DOWN_QPOS_HOME[7:].copy()
        self.reset()

    def default_state(self) -> np.ndarray:
        return np.concatenate([LEAP_DOWN_QPOS_HOME, np.zeros(22)])


CALTECH_LEAP_QPOS_HOME = np.array(
    [0.11, 0.005, 0.04, 1.0, 0.0, 0.0, 0.0, 0.5, -0.75, 0.75, 0.25, 0.5, 0.0, 0.75, 0.25, 0.5, 0.75, 0.75, 0.25, 0.65, 0.9, 0.75, 0.6]
)  # judo/tasks/caltech_leap_cube.py:13-21


@dataclass
class CaltechLeapCubeConfig(LeapCubeConfig):
    pass


class CaltechLeapCube(LeapCube):
    """The Caltech LEAP hand (judo/tasks/caltech_leap_cube.py:31-51): the same 16-joint hand built from primitive geoms, palm up at the origin, cone
    `impratio` 1, the cube above the grasp site; sensordata is the 16 joint positions, the cube position in the grasp-site frame and the cube
    orientation relative to the goal body.  Runs on the leap_cube kernels (generation 3) with its own model constants."""

    name = "caltech_leap_cube"
    config_t = CaltechLeapCubeConfig

    def __init__(self) -> None:
        Task.__init__(self)
        self.goal_pos = np.array([0.11, 0.005, 0.03])
        self.goal_quat = np.array([1.0, 0.0, 0.0, 0.0])
        self.qpos_home = CALTECH_LEAP_QPOS_HOME
        self.reset_command = CALTECH_LEAP_QPOS_HOME[7:].copy()
        self.reset()

    def default_state(self) -> np.ndarray:
        return np.concatenate([CALTECH_LEAP_QPOS_HOME, np.zeros(22)])


# ------------------------------------------------------------------------------------------------ fr3_pick
FR3_QPOS_HOME = np.array([0.7, 0, 0.02, 1, 0, 0, 0, 0, -0.7854, 0.0, -2.3562, 0.0, 1.5708, 0.7854, 0.04, 0.04])  # fr3_pick.py:16-22


class Phase(Enum):
    LIFT = 0
    MOVE = 1
    PLACE = 2
    HOMING = 3


@dataclass
class LiftConfig:
    w_lift_close: float = 1.0
    w_lift_height: float = 10.0


@dataclass
class MoveConfig:
    w_move_goal: float = 1.0
    w_move_close: float = 10.0


@dataclass
class PlaceConfig:
    w_place_table: float = 1.0
    w_place_goal: float = 1.0


@dataclass
class GlobalConfig:
    w_upright: float = 0.25
    w_coll: float = 0.1
    w_qvel: float = 0.005
    w_open: float = 2.0


@dataclass
class FR3PickConfig(TaskConfig):
    lift_weights: LiftConfig = field(default_factory=LiftConfig)
    move_weights: MoveConfig = field(default_factory=MoveConfig)
    place_weights: PlaceConfig = field(default_factory=PlaceConfig)
    global_weights: GlobalConfig = field(default_factory=GlobalConfig)
    goal_pos: np.ndarray = field(default_factory=lambda: np.array([0.6, 0.4]))
    goal_radius: float = 0.05
    pick_height: float = 0.3


class FR3Pick(Task[FR3PickConfig]):
    name = "fr3_pick"
    config_t = FR3PickConfig

    def __init__(self) -> None:
        super().__init__()
        self.reset_command = np.array([0, 0, 0, -1.57079, 0, 1.57079, -0.7853, 0.0])
        self._phase = Phase.LIFT
        self.reset()

    @property
    def phase(self) -> int:
        return self._phase.value

    @phase.setter
    def phase(self, v) -> None:
        self._phase = v if isinstance(v, Phase) else Phase(int(v))

    def task_params(self, system_metadata=None) -> np.ndarray:
        c = self.config
        return np.array(
            [c.lift_weights.w_lift_close, c.lift_weights.w_lift_height, c.move_weights.w_move_goal, c.move_weights.w_move_close,
             c.place_weights.w_place_table, c.place_weights.w_place_goal, c.global_weights.w_upright, c.global_weights.w_coll,
             c.global_weights.w_qvel, c.global_weights.w_open, c.goal_pos[0], c.goal_pos[1], c.pick_height, *FR3_QPOS_HOME[7:16]],
            dtype=np.float32,
        )

    def optimizer_warm_start(self) -> np.ndarray:
        return self.reset_command.copy()

    def pre_rollout(self, curr_state: np.ndarray) -> None:
        """Phase decision from the object's height and xy distance to the goal (fr3_pick.py:191-223)."""
        obj = curr_state[0:3]
        in_air = obj[2] > 0.02 + 1e-3
        in_goal = np.linalg.norm(obj[:2] - self.config.goal_pos) <= self.config.goal_radius
        phase = Phase.LIFT
        if in_air:
            phase = Phase.MOVE
        if in_goal and in_air:
            phase = Phase.PLACE
        if in_goal and obj[2] <= 0.02 + 1e-3:
            phase = Phase.HOMING
        self._phase = phase

    def reset(self) -> None:
        self.data.qpos = FR3_QPOS_HOME.copy()
        self.data.qvel = np.zeros(self.nv)

    def default_state(self) -> np.ndarray:
        return np.concatenate([FR3_QPOS_HOME, np.zeros(15)])


_registered_tasks: dict[str, tuple[type, type]] = {
    Cartpole.name: (Cartpole, CartpoleConfig),
    CylinderPush.name: (CylinderPush, CylinderPushConfig),
    LeapCube.name: (LeapCube, LeapCubeConfig),
    LeapCubeDown.name: (LeapCubeDown, LeapCubeDownConfig),
    CaltechLeapCube.name: (CaltechLeapCube, CaltechLeapCubeConfig),
    FR3Pick.name: (FR3Pick, FR3PickConfig),
}


def get_registered_tasks() -> dict[str, tuple[type, type]]:
    return _registered_tasks


def register_task(name: str, task_type: type, task_config_type: type) -> None:
    """judo/tasks/__init__.py:45."""
    _registered_tasks[name] = (task_type, task_config_type)


from judo_amd import spot_tasks as _spot_tasks  # noqa: E402,F401  (registers spot_base / spot_navigate)
