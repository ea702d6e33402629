"""KickEnv with the reference's surface (bez_isaacgym/tasks/kick_env.py) over the HIP simulator.

What stays drop-in: ctor `(cfg, sim_device, graphics_device_id, headless)` (kick_env.py:46); attributes the
reference's tests / controllers read -- actions, device, dt, dof_pos_limits_lower/upper, default_dof_pos,
root_pos_bez, root_orient_bez, num_envs, reset_buf, root_states, dof_state, rigid_body ... (kick_env.py:126-238);
hooks pre_physics_step (:410), post_physics_step (:426), reset_idx (:779), compute_observations (:749),
compute_reward (:584).  What changed: every hook is a call into libbez_sim.so; step() is the fused kernel.
"""
import enum
import json
import os

import numpy as np
import torch

from .. import abi
from ..sim import BezSim
from .base.vec_task import VecTask


class Joints(enum.IntEnum):  # kick_env.py:23-41
    HEAD_1 = 0
    HEAD_2 = 1
    LEFT_ARM_1 = 2
    LEFT_ARM_2 = 3
    LEFT_LEG_1 = 4
    LEFT_LEG_2 = 5
    LEFT_LEG_3 = 6
    LEFT_LEG_4 = 7
    LEFT_LEG_5 = 8
    LEFT_LEG_6 = 9
    RIGHT_ARM_1 = 10
    RIGHT_ARM_2 = 11
    RIGHT_LEG_1 = 12
    RIGHT_LEG_2 = 13
    RIGHT_LEG_3 = 14
    RIGHT_LEG_4 = 15
    RIGHT_LEG_5 = 16
    RIGHT_LEG_6 = 17


_MODEL = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "model", "bez_model.json")))


class KickEnv(VecTask):
    """bez_kick.  WalkEnv / OrientEnv (tasks/walk_env.py, tasks/orient_env.py) derive from it: same robot, same tensor API,
    no ball actor, 52 observations, their own reward / reset logic inside the kernel (BezSimConfig.task)."""
    TASK = "bez_kick"
    HAS_BALL = True

    def __init__(self, cfg, sim_device, graphics_device_id, headless):
        self.cfg = cfg
        self.randomization_params = self.cfg["task"]["randomization_params"]
        self.randomize = self.cfg["task"]["randomize"]
        env = self.cfg["env"]
        self.plane_static_friction = env["plane"]["staticFriction"]
        self.plane_dynamic_friction = env["plane"]["dynamicFriction"]
        self.plane_restitution = env["plane"]["restitution"]
        self.bez_init_state = env["bezInitState"]["pos"] + env["bezInitState"]["rot"] + \
            env["bezInitState"]["vLinear"] + env["bezInitState"]["vAngular"]
        if self.HAS_BALL:
            self.ball_init_state = env["ballInitState"]["pos"] + env["ballInitState"]["rot"] + \
                env["ballInitState"]["vLinear"] + env["ballInitState"]["vAngular"]
        goal = env["goalState"]["goal"]
        self.cleats = bool(env["asset"]["cleats"])  # True -> soccerbot_stl_sensor.urdf: 29 bodies, per-cleat contact rows
        # asset.stl: False -> soccerbot_box.urdf / soccerbot_box_sensor.urdf (kick_env.py:272-276): the same robot with box
        # collision shapes on the torso / head / arms (BEZ_FLAG_BOX_ASSET, set by abi.config_from_task_cfg)
        self.box_asset = not bool(env["asset"]["stl"])
        self.debug_rewards = env["debug"]["rewards"]
        self.named_default_joint_angles = env["readyJointAngles"]
        self.max_episode_length_s = env["learn"]["episodeLength_s"]
        self.Kp = env["control"]["stiffness"]
        self.Kd = env["control"]["damping"]
        self.orn_dim, self.imu_dim, self.feet_dim, self.dof_dim, self.rnn_dim = 2, 6, 8, 18, 1
        self.ball_dim = 2 if self.HAS_BALL else 0
        self.imu_max_ang_vel = 8.7266
        self.imu_max_lin_acc = 2. * 9.81
        self.MX_28_velocity = 2 * np.pi
        self.cfg["env"]["numObservations"] = self.dof_dim * 2 + self.imu_dim + self.orn_dim + self.feet_dim + self.ball_dim  # 54
        self.cfg["env"]["numActions"] = self.dof_dim
        self.strict_reference_quirks = bool(self.cfg.get("strict_reference_quirks", True))

        super().__init__(config=self.cfg, sim_device=sim_device, graphics_device_id=graphics_device_id, headless=headless)

        self.dt = self.cfg["sim"]["dt"]
        self.max_episode_length = int(self.max_episode_length_s / self.dt + 0.5)
        dev = self.device
        n = self.num_envs
        self._goal_cfg = torch.tensor([goal], device=dev, dtype=torch.float32).repeat((n, 1))
        self.bez_init_xy = torch.tensor(self.bez_init_state[0:2], device=dev, dtype=torch.float32)
        actors = [self.bez_init_state]
        if self.HAS_BALL:
            self.ball_init = torch.tensor([self.ball_init_state[0:2]], device=dev, dtype=torch.float32).repeat((n, 1))
            actors.append(self.ball_init_state)
        self.initial_root_states = torch.tensor(actors, device=dev, dtype=torch.float32).repeat((n, 1))
        self.initial_root_states[:, 7:13] = 0
        self.default_dof_pos = torch.tensor(_MODEL["dof_default"], device=dev, dtype=torch.float32).repeat((n, 1))
        for i, name in enumerate(self.dof_names):
            self.default_dof_pos[:, i] = float(self.named_default_joint_angles[name])
        self.num_dofs = self.num_dof
        self.actions = torch.zeros(n, self.num_actions, dtype=torch.float, device=dev)
        self.gravity_vec = torch.tensor([0., 0., -1.], device=dev).repeat((n, 1))
        self.up_vec = torch.tensor([0., 0., 1.], device=dev).repeat((n, 1))
        self._stale = True
        # (reset_idx(all envs), kick_env.py:238, already happened inside bez_sim_create)

    # ---- sim construction (kick_env.py:240-408)
    def create_sim(self):
        self.up_axis_idx = 2
        rank_offset = int(self.cfg.get("env_id_offset", 0))
        seed = int(self.cfg.get("seed", 42))
        sim_cfg = abi.config_from_task_cfg(self.cfg, seed=seed, env_id_offset=rank_offset,
                                           strict_reference_quirks=self.strict_reference_quirks, task=self.TASK)
        self.sim_cfg = sim_cfg
        self.sim = BezSim(sim_cfg, self.device_id)
        self.num_dof = 18
        self.num_bodies = 29 if self.cleats else 21
        self.num_joints = self.num_bodies - 1
        self.dof_names = list(_MODEL["dof_names"])
        na = self.sim.num_actors
        self.bez_indices = torch.arange(0, self.num_envs * na, na, device=self.device, dtype=torch.long)
        if self.HAS_BALL:
            self.ball_indices = self.bez_indices + 1
        self.dof_pos_limits_lower = torch.tensor(_MODEL["dof_lower"], device=self.device, dtype=torch.float32)
        self.dof_pos_limits_upper = torch.tensor(_MODEL["dof_upper"], device=self.device, dtype=torch.float32)
        self.dof_vel_limits_upper = torch.full((18, 1), self.MX_28_velocity, device=self.device)
        self.dof_vel_limits_lower = -self.dof_vel_limits_upper
        self.start_rotation = torch.tensor([0., 0., 0., 1.], device=self.device)
        if self.randomize:
            self.apply_randomizations(self.randomization_params)

    # ---- lean stepping: a consumer that reads only obs / reward / reset (a PPO rollout) may tell the simulator not to keep the
    # net-contact-force rows, self.feet and prev_lin_vel current (BEZ_FLAG_LEAN_STEP: 308 of the 912 bytes an env-step writes)
    def set_lean(self, on):
        f = int(self.sim.cfg.flags)
        f = (f | abi.FLAG_LEAN_STEP) if on else (f & ~abi.FLAG_LEAN_STEP)
        self.sim.cfg.flags = f
        self.sim.set_flags(f)
        # the kernels honour the flag only together with BEZ_FLAG_IMU_PREV_ALIAS (prev_lin_vel is then never read back,
        # include/bez_sim.h); without it every tensor stays current and the accessors below must not refuse
        self._lean = bool(on) and bool(f & abi.FLAG_IMU_PREV_ALIAS)

    def _not_lean(self, what):
        if getattr(self, "_lean", False):
            raise RuntimeError("%s is not kept current while the env steps lean (BEZ_FLAG_LEAN_STEP): call env.set_lean(False) and "
                               "step again before reading it" % what)

    # ---- Isaac-layout tensors and the reference's views of them (kick_env.py:143-196), refreshed lazily
    def _refresh_all(self):
        if self._stale:
            for t in (abi.TENSOR_ROOT_STATE, abi.TENSOR_DOF_STATE, abi.TENSOR_RIGID_BODY_STATE, abi.TENSOR_NET_CONTACT_FORCE):
                self.sim.refresh(t)
            self._stale = False

    @property
    def root_states(self):
        self._refresh_all(); return self.sim.tensor(abi.TENSOR_ROOT_STATE)

    @property
    def dof_state(self):
        self._refresh_all(); return self.sim.tensor(abi.TENSOR_DOF_STATE)

    @property
    def rigid_body(self):
        self._refresh_all(); return self.sim.tensor(abi.TENSOR_RIGID_BODY_STATE)

    @property
    def net_contact_forces(self):
        self._not_lean("net_contact_forces")
        self._refresh_all(); return self.sim.tensor(abi.TENSOR_NET_CONTACT_FORCE)

    dof_pos_bez = property(lambda s: s.dof_state.view(s.num_envs, 18, 2)[..., 0])
    dof_vel_bez = property(lambda s: s.dof_state.view(s.num_envs, 18, 2)[..., 1])
    root_pos_bez = property(lambda s: s.root_states.view(s.num_envs, -1, 13)[..., 0, 0:3])
    root_orient_bez = property(lambda s: s.rigid_body.view(s.num_envs, -1, 13)[..., 1, 3:7])
    root_vel_bez = property(lambda s: s.rigid_body.view(s.num_envs, -1, 13)[..., 1, 7:10])
    root_ang_bez = property(lambda s: s.rigid_body.view(s.num_envs, -1, 13)[..., 1, 10:13])
    root_pos_ball = property(lambda s: s.root_states.view(s.num_envs, -1, 13)[..., 1, 0:3])
    root_orient_ball = property(lambda s: s.root_states.view(s.num_envs, -1, 13)[..., 1, 3:7])
    root_vel_ball = property(lambda s: s.root_states.view(s.num_envs, -1, 13)[..., 1, 7:10])
    # kick_env.py:187-196: per-cleat rows with the cleats asset, the two foot rows otherwise
    left_contact_forces = property(lambda s: s.net_contact_forces.view(s.num_envs, -1, 3)[..., 13:17, 0:3])
    right_contact_forces = property(lambda s: s.net_contact_forces.view(s.num_envs, -1, 3)[..., 25:29, 0:3])
    left_foot_contact_forces = property(lambda s: s.net_contact_forces.view(s.num_envs, -1, 3)[..., 12, 0:3])
    right_foot_contact_forces = property(lambda s: s.net_contact_forces.view(s.num_envs, -1, 3)[..., 24 if s.cleats else 20, 0:3])

    @property
    def goal(self):
        """(N,2) goal: constant for bez_kick; bez_walk / bez_orient redraw it inside reset_idx (walk_env.py:570-575)."""
        return self._goal_cfg if self.HAS_BALL else self.sim.refresh(abi.TENSOR_GOAL)
    prev_lin_vel = property(lambda s: (s._not_lean("prev_lin_vel"), s.sim.refresh(abi.TENSOR_PREV_LIN_VEL))[1])
    feet = property(lambda s: (s._not_lean("feet"), s.sim.refresh(abi.TENSOR_FEET))[1])

    # ---- step
    @property
    def actions(self):
        """The reference keeps `self.actions` = clamped actions with the head zeroed (vec_task.py:317, kick_env.py:411-414).
        The kernel does that clamp itself; the Python-visible copy is materialised only when somebody reads it."""
        if self._raw_actions is not None:
            a = torch.clamp(self._raw_actions, -self.clip_actions, self.clip_actions)
            a[..., 0:2] = 0.0
            self._actions, self._raw_actions = a, None
        return self._actions

    @actions.setter
    def actions(self, value):
        self._actions, self._raw_actions = value, None

    @property
    def graph_safe(self):
        """True: step() never syncs with the host nor reads host-side state that changes between steps -- domain randomisation
        resamples inside the simulator (bez_sim_set_randomization), the bez_walk / bez_orient goal is drawn from a device-resident
        counter -- so the PPO loop may capture it into a HIP graph."""
        return True

    def _fused_step(self, actions):
        if self.control_freq_inv != 1:
            # vec_task.py:322-324 loops gym.simulate controlFrequencyInv times per env step: the fused kernel is one simulate per
            # step, so any other value takes the split entry points (pre_physics, simulate x k, post_physics)
            return VecTask._fused_step(self, actions)
        self._raw_actions = actions  # borrowed until the next step (see `actions`)
        self.sim.step(actions)  # with randomize: True the kernel in front of the step counts randomize_buf and redraws at reset time
        self._stale = True

    def pre_physics_step(self, actions):
        self.actions = actions.clone().to(self.device)
        self.sim.pre_physics(actions.to(self.device, torch.float32).contiguous())
        self.actions[..., 0:2] = 0.0

    def post_physics_step(self):
        self.sim.post_physics()
        if not self.randomize:
            self.randomize_buf += 1   # kick_env.py:430; with randomize: True the simulator counts it (and clears it at a redraw)
        self._stale = True

    def compute_observations(self):
        self.sim.observe_reward()
        self._stale = True

    def compute_reward(self, actions=None):
        pass  # produced together with the observations by the same kernel

    def reset_idx(self, env_ids):
        if self.randomize:
            self.apply_randomizations(self.randomization_params)
        self.sim.reset_indexed(env_ids.to(self.device, torch.int32).contiguous())
        self._stale = True
