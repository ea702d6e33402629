"""numpy-in / numpy-out adapter over the HIP simulator's C ABI with the same interface as
oracle.bez_oracle.Oracle, so that golden checks and oracle-parity tests read identically."""
import numpy as np
import torch

from bez_isaacgym_amd import abi
from bez_isaacgym_amd.sim import BezSim


class SimAdapter:
    def __init__(self, cfg=None, num_envs=64):
        self.cfg = cfg if cfg is not None else abi.default_config(num_envs)
        self.sim = BezSim(self.cfg, 0)
        self.n = self.sim.num_envs
        self.dev = self.sim.device
        self.nact, self.nbe, self.nobs = self.sim.num_actors, self.sim.num_bodies, self.sim.num_obs
        self._robot_ids = torch.arange(0, self.n * self.nact, self.nact, dtype=torch.int32, device=self.dev)
        self._all_ids = torch.arange(0, self.n * self.nact, dtype=torch.int32, device=self.dev)

    def _t(self, a, dtype=torch.float32):
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(self.dev).contiguous()

    def _get(self, which):
        t = self.sim.refresh(which)
        torch.cuda.synchronize()
        return t.detach().cpu().numpy().copy()

    root_states = property(lambda s: s._get(abi.TENSOR_ROOT_STATE))
    dof_state = property(lambda s: s._get(abi.TENSOR_DOF_STATE))
    rigid_body_states = property(lambda s: s._get(abi.TENSOR_RIGID_BODY_STATE))
    contact_forces = property(lambda s: s._get(abi.TENSOR_NET_CONTACT_FORCE))
    targets = property(lambda s: s._get(abi.TENSOR_DOF_TARGET))
    goal = property(lambda s: s._get(abi.TENSOR_GOAL))
    prev_lin_vel = property(lambda s: s._get(abi.TENSOR_PREV_LIN_VEL))
    obs = property(lambda s: s._get(abi.TENSOR_OBS))
    feet = property(lambda s: s._get(abi.TENSOR_FEET))
    rew = property(lambda s: s._get(abi.TENSOR_REW))
    reset_buf = property(lambda s: s._get(abi.TENSOR_RESET))
    progress_buf = property(lambda s: s._get(abi.TENSOR_PROGRESS))
    timeout_buf = property(lambda s: s._get(abi.TENSOR_TIMEOUT))
    randomize_buf = property(lambda s: s.sim.tensor(abi.TENSOR_RANDOMIZE_BUF).detach().cpu().numpy().copy())
    dr_noise = property(lambda s: s.sim.tensor(abi.TENSOR_DR_NOISE).detach().cpu().numpy().copy())

    def set_randomize(self, a): self.sim.tensor(abi.TENSOR_RANDOMIZE_BUF).copy_(self._t(a, torch.int64))
    def set_randomization(self, dr): self.sim.set_randomization(dr)
    def get_env_params(self, param): return self.sim.get_env_params(param).detach().cpu().numpy().copy()

    def set_root_states(self, a): self.sim.set_actor_root_state_tensor_indexed(self._t(a).reshape(-1), self._all_ids)
    def set_dof_state(self, a): self.sim.set_dof_state_tensor_indexed(self._t(a).reshape(-1), self._robot_ids)
    def set_contact_forces(self, a): self.sim.set_net_contact_force_tensor(self._t(a).reshape(-1))
    def set_targets(self, a): self.sim.set_dof_position_target_tensor(self._t(a).reshape(-1))
    def set_goal(self, a): self.sim.set_goal_tensor(self._t(a).reshape(-1))
    def set_prev_lin_vel(self, a): self.sim.set_prev_lin_vel_tensor(self._t(a).reshape(-1))
    def set_reset(self, a): self.sim.tensor(abi.TENSOR_RESET).copy_(self._t(a, torch.int64))
    def set_progress(self, a): self.sim.tensor(abi.TENSOR_PROGRESS).copy_(self._t(a, torch.int64))
    def set_flags(self, f): self.sim.set_flags(f)
    def set_obs_calls(self, n): self.sim.set_obs_calls(n)
    def seed(self, s): self.sim.seed(s)

    def set_env_params(self, param, values):
        self.sim.set_env_params(param, None if values is None else self._t(values).reshape(-1))

    def pre_physics(self, actions): self.sim.pre_physics(self._t(actions).reshape(-1))
    def simulate(self): self.sim.simulate()
    def post_physics(self): self.sim.post_physics()
    def step(self, actions): self.sim.step(self._t(actions).reshape(-1))

    def observe_reward(self): self.sim.observe_reward()

    def reset_idx(self, ids):
        self.sim.reset_indexed(torch.as_tensor(np.asarray(ids), dtype=torch.int32).to(self.dev))
