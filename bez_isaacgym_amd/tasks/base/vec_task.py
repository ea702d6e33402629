"""Env / VecTask base classes with the reference's public surface (bez_isaacgym/tasks/base/vec_task.py),
re-implemented over the HIP simulator: there is no gymapi, `gym.simulate` is libbez_sim.so.

Kept from the reference: constructor signature (vec_task.py:51,150), device parsing (:61-73), spaces
(:92-95), buffer names/dtypes/initial values (:226-249), step()/reset() return contract (:303-377),
zero_actions (:351-359), get_state (:287-289), properties (:122-145), domain-randomisation entry point
apply_randomizations (:505-725, here: per-env parameter arrays pushed to the kernel).
"""
import abc
from typing import Any, Dict, Tuple

import numpy as np
import torch


class Box:
    """Minimal stand-in for gym.spaces.Box (gym is not a dependency): low/high/shape/dtype."""

    def __init__(self, low, high, dtype=np.float32):
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)

    def __repr__(self):
        return "Box(%s, %s, %s, %s)" % (self.low.min(), self.high.max(), self.shape, self.dtype)


class Env(abc.ABC):
    def __init__(self, config: Dict[str, Any], sim_device: str, graphics_device_id: int, headless: bool):
        split_device = sim_device.split(":")
        self.device_type = split_device[0]
        self.device_id = int(split_device[1]) if len(split_device) > 1 else 0

        # vec_task.py:65-71.  This build has no CPU pipeline: the simulator is HIP-only.
        self.device = "cpu"
        if config["sim"]["use_gpu_pipeline"]:
            if self.device_type.lower() in ("cuda", "gpu"):
                self.device = "cuda" + ":" + str(self.device_id)
            else:
                print("GPU Pipeline can only be used with GPU simulation. Forcing CPU Pipeline.")
                config["sim"]["use_gpu_pipeline"] = False

        self.rl_device = config.get("rl_device", "cuda:0")
        self.headless = headless
        self.graphics_device_id = graphics_device_id
        if not config.get("enableCameraSensors", False) and self.headless:
            self.graphics_device_id = -1

        self.num_environments = config["env"]["numEnvs"]
        self.num_agents = config["env"].get("numAgents", 1)
        self.num_observations = config["env"]["numObservations"]
        self.num_states = config["env"].get("numStates", 0)
        self.num_actions = config["env"]["numActions"]
        self.control_freq_inv = config["env"].get("controlFrequencyInv", 1)

        self.obs_space = Box(np.ones(self.num_obs) * -np.inf, np.ones(self.num_obs) * np.inf)
        self.state_space = Box(np.ones(self.num_states) * -np.inf, np.ones(self.num_states) * np.inf)
        self.act_space = Box(np.ones(self.num_actions) * -1., np.ones(self.num_actions) * 1.)

        self.clip_obs = config["env"].get("clipObservations", np.inf)
        self.clip_actions = config["env"].get("clipActions", np.inf)

    @abc.abstractmethod
    def allocate_buffers(self): ...

    @abc.abstractmethod
    def step(self, actions: torch.Tensor) -> Tuple[Dict[str, torch.Tensor], torch.Tensor, torch.Tensor, Dict[str, Any]]: ...

    @abc.abstractmethod
    def reset(self) -> Dict[str, torch.Tensor]: ...

    @property
    def observation_space(self):
        return self.obs_space

    @property
    def action_space(self):
        return self.act_space

    @property
    def num_envs(self) -> int:
        return self.num_environments

    @property
    def num_acts(self) -> int:
        return self.num_actions

    @property
    def num_obs(self) -> int:
        return self.num_observations


MASS_DRAW_TAG = 0x4D415353  # "MASS": Philox counter word of the setup-only mass scale draw


class VecTask(Env):
    """Subclasses create `self.sim` (a bez_isaacgym_amd.sim.BezSim) in create_sim()."""

    def __init__(self, config, sim_device, graphics_device_id, headless):
        super().__init__(config, sim_device, graphics_device_id, headless)
        if self.cfg["physics_engine"] not in ("physx", "flex"):  # vec_task.py:162-168
            raise ValueError(f"Invalid physics engine backend: {self.cfg['physics_engine']}")
        if self.cfg["sim"]["up_axis"] not in ["z", "y"]:        # vec_task.py:421-424
            msg = f"Invalid physics up-axis: {self.cfg['sim']['up_axis']}"
            print(msg)
            raise ValueError(msg)
        if self.device == "cpu":
            raise RuntimeError("bez_isaacgym_amd is HIP-only: run with sim_device=cuda:<k> pipeline=gpu "
                               "(the CPU restatement lives in oracle/ and is test infrastructure)")
        self.first_randomization = True
        self.dr_randomizations = {}
        self.last_step = -1
        self.last_rand_step = -1
        self.viewer = None
        self.enable_viewer_sync = True
        self.sim_initialized = False
        self.create_sim()
        self.sim_initialized = True
        self.allocate_buffers()
        self.obs_dict = {}

    def allocate_buffers(self):
        """Same names / dtypes as vec_task.py:226-249; obs/rew/reset/progress/timeout are zero-copy views of
        the simulator's own buffers (the fused kernel writes them in place)."""
        from ... import abi
        s = self.sim
        self.obs_buf = s.tensor(abi.TENSOR_OBS)
        self.states_buf = torch.zeros((self.num_envs, self.num_states), device=self.device, dtype=torch.float)
        self.rew_buf = s.tensor(abi.TENSOR_REW)
        self.reset_buf = s.tensor(abi.TENSOR_RESET)
        self.timeout_buf = s.tensor(abi.TENSOR_TIMEOUT)
        self.progress_buf = s.tensor(abi.TENSOR_PROGRESS)
        self.randomize_buf = s.tensor(abi.TENSOR_RANDOMIZE_BUF)
        self.extras = {}

    def get_state(self):
        return torch.clamp(self.states_buf, -self.clip_obs, self.clip_obs).to(self.rl_device)

    @abc.abstractmethod
    def pre_physics_step(self, actions: torch.Tensor): ...

    @abc.abstractmethod
    def post_physics_step(self): ...

    def step(self, actions: torch.Tensor):
        """vec_task.py:303-349.  With no Python-side hooks active this is ONE kernel launch
        (bez_sim_step: clamp, PD targets, physics, bookkeeping, reset, obs, reward)."""
        if self.dr_randomizations.get('actions', None) and not self.external_action_noise:
            actions = self.dr_randomizations['actions']['noise_lambda'](actions)
        actions = actions.to(self.device, torch.float32).contiguous()
        self._fused_step(actions)
        if self.dr_randomizations.get('observations', None):
            out = self.dr_randomizations['observations']['noise_lambda'](self.obs_buf)
            if out.data_ptr() != self.obs_buf.data_ptr():
                self.obs_buf.copy_(out)
        self.extras["time_outs"] = self.timeout_buf.to(self.rl_device)
        self.obs_dict["obs"] = self._clipped_obs()
        if self.num_states > 0:
            self.obs_dict["states"] = self.get_state()
        return self.obs_dict, self.rew_buf.to(self.rl_device), self.reset_buf.to(self.rl_device), self.extras

    def _clipped_obs(self):
        """vec_task.py:347: clamp(obs_buf, +-clipObservations).to(rl_device).  With the default clipObservations = inf the clamp
        is the identity: the simulator's own buffer is returned (zero-copy, valid until the next step()) instead of a
        full-tensor copy per control step.  A consumer that keeps obs_dict["obs"] across step() calls sets
        env.copyObservations: True in the task config to get the reference's fresh tensor per step."""
        if np.isinf(self.clip_obs) and not self.cfg["env"].get("copyObservations", False):
            return self.obs_buf.to(self.rl_device)
        return torch.clamp(self.obs_buf, -self.clip_obs, self.clip_obs).to(self.rl_device)

    def _fused_step(self, actions):
        """Default: the split path through the subclass hooks (vec_task.py:317-335)."""
        action_tensor = torch.clamp(actions, -self.clip_actions, self.clip_actions)
        self.pre_physics_step(action_tensor)
        for _ in range(self.control_freq_inv):
            self.render()
            self.sim.simulate()
        self.post_physics_step()

    def zero_actions(self) -> torch.Tensor:
        return torch.zeros([self.num_envs, self.num_actions], dtype=torch.float32, device=self.rl_device)

    def reset(self):
        """vec_task.py:361-377: one step with zero actions."""
        self.step(self.zero_actions())
        self.obs_dict["obs"] = self._clipped_obs()
        if self.num_states > 0:
            self.obs_dict["states"] = self.get_state()
        return self.obs_dict

    def render(self):
        """Headless only (viewer is out of scope): no-op."""
        return None

    def get_number_of_agents(self):
        return self.num_agents

    # ---- domain randomisation (vec_task.py:505-725), device-side: bez_sim_set_randomization
    # ---- hooks for a trainer that drives the randomised env at full speed (ppo/a2c_continuous.py): both default to "the env does it"
    external_action_noise = False   # True: the caller adds the action noise itself (action_noise_source), step() must not add it again

    def action_noise_source(self):
        """(snapshot pointer, seed, env id offset) of the action-noise lambda for a consumer that adds it in its own launch, or None"""
        return self.sim.action_noise_source() if self.dr_randomizations.get('actions', None) else None

    def dr_prelaunch(self):
        """The coming step's randomisation kernel now, on the current stream (overlappable with the policy's forward pass); no-op
        without a device-side randomisation."""
        if not self.first_randomization and self.randomize:
            self.sim.dr_prelaunch()

    def dr_step_args(self):
        """The coming step's randomisation as an argument block for a launch that carries it (sim.dr_step_args); None without one."""
        return self.sim.dr_step_args() if (not self.first_randomization and self.randomize) else None

    def apply_randomizations(self, dr_params):
        """The reference calls this from reset_idx on every step in which some env resets (kick_env.py:781-782); what it does
        there -- per-env redraws at reset time once `frequency` steps have passed, gravity and the noise parameters on the same
        clock, all scaled by the linear schedule -- now happens inside the simulator, in a small kernel in front of the step
        kernel (include/bez_sim.h: bez_sim_set_randomization), keyed by (seed, global env id, episode).  The first call hands
        the parameters over (and randomises every env at frame 0, first_randomization); later calls have nothing left to do,
        so the step never syncs with the host and stays HIP-graph capturable with randomize: True."""
        from ... import abi
        if not self.first_randomization:
            return
        self.sim.set_randomization(abi.dr_config_from_params(dr_params))
        self.randomize_buf = self.sim.tensor(abi.TENSOR_RANDOMIZE_BUF)     # the kernel counts it (kick_env.py:430)
        # vec_task.py:544-618 noise lambdas (gaussian additive, range_correlated absent in bez_kick.yaml): one launch each, the mean /
        # std (BEZ_TENSOR_DR_NOISE, moved by the schedule) are read on the device.  The caller's action tensor is not modified.
        sim = self.sim
        if "observations" in dr_params:
            # BEZ_FLAG_OBS_NOISE_IN_STEP: the step kernel adds this noise on its way out (one launch less per step); on the simulator's
            # own observation buffer the call below is then a no-op inside the library, any other tensor still gets a launch
            sim.set_flags(int(sim.cfg.flags) | abi.FLAG_OBS_NOISE_IN_STEP)
            sim.cfg.flags = int(sim.cfg.flags) | abi.FLAG_OBS_NOISE_IN_STEP
            self.dr_randomizations["observations"] = {"noise_lambda": lambda t: sim.add_dr_noise(t if t.is_contiguous() else t.contiguous(), 0)}
        if "actions" in dr_params:
            self._noisy_actions = torch.empty(self.num_envs, self.num_actions, device=self.device, dtype=torch.float32)  # persistent: graph-safe

            def act_noise(t):
                t = t.to(self.device, torch.float32)
                return sim.add_dr_noise(t if t.is_contiguous() else t.contiguous(), 1, out=self._noisy_actions.view(t.shape))
            self.dr_randomizations["actions"] = {"noise_lambda": act_noise}
        # rigid_shape_properties.restitution (bez_kick.yaml:187-192) SCALES the asset's restitution, which is 0 (plane
        # restitution 0, bez_kick.yaml:16; asset default 0 [ext]): 0 * U(0, 0.7) = 0 -- nothing to randomise.
        # rigid_body_properties.mass is setup_only (bez_kick.yaml:175): drawn once, here, at frame 0 -- where its linear schedule
        # still interpolates to "no randomisation" (scale 1); without a schedule it is a real one-time draw.
        rbp = (((dr_params.get("actor_params") or {}).get("bez") or {}).get("rigid_body_properties") or {})
        if "mass" in rbp and rbp["mass"].get("schedule") != "linear":
            # one draw per (GLOBAL env id, link) from the simulator's own counter-based generator: the same env gets the same
            # masses whatever the number of ranks (every other per-env draw is keyed that way, include/bez_sim.h)
            from ...utils.utils import per_env_uniform
            lo, hi = (float(v) for v in rbp["mass"]["range"])
            off = int(self.cfg.get("env_id_offset", 0))
            u = per_env_uniform(int(self.cfg.get("seed", 42)), range(off, off + self.num_envs), MASS_DRAW_TAG, 19)
            self.sim.set_env_params(abi.PARAM_MASS_SCALE, torch.from_numpy(u * (hi - lo) + lo).to(self.device).contiguous())
        self.first_randomization = False
