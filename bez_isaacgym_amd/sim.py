"""Python binding of the C ABI (include/bez_sim.h) -- thin ctypes calls, torch only for device memory
and the current HIP stream.  There is NO CPU fallback: if libbez_sim.so is missing or no GPU is
visible, construction raises."""
import ctypes as C
import os

import torch

from . import abi
from .build import lib_path

_LIB = None


class BezSimError(RuntimeError):
    pass


def load_library():
    """dlopen libbez_sim.so and declare every entry point of include/bez_sim.h."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise BezSimError("libbez_sim.so not built (%s): run `python -m bez_isaacgym_amd.build` -- "
                          "the HIP extension is required, there is no fallback path" % path)
    lib = C.CDLL(path)
    vp, i32, i64, u32, u64, fp = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_void_p
    sigs = {
        "bez_sim_default_config": (C.c_int, [C.POINTER(abi.BezSimConfig), i32]),
        "bez_sim_create": (C.c_int, [C.POINTER(abi.BezSimConfig), C.c_int, C.POINTER(vp)]),
        "bez_sim_destroy": (C.c_int, [vp]),
        "bez_sim_last_error": (C.c_char_p, [vp]),
        "bez_sim_get_tensor": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(i64), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "bez_sim_refresh_tensor": (C.c_int, [vp, C.c_int, vp]),
        "bez_sim_set_actor_root_state_tensor_indexed": (C.c_int, [vp, fp, vp, i32, vp]),
        "bez_sim_set_dof_state_tensor_indexed": (C.c_int, [vp, fp, vp, i32, vp]),
        "bez_sim_set_dof_position_target_tensor": (C.c_int, [vp, fp, vp]),
        "bez_sim_set_dof_position_target_tensor_indexed": (C.c_int, [vp, fp, vp, i32, vp]),
        "bez_sim_set_net_contact_force_tensor": (C.c_int, [vp, fp, vp]),
        "bez_sim_set_prev_lin_vel_tensor": (C.c_int, [vp, fp, vp]),
        "bez_sim_set_goal_tensor": (C.c_int, [vp, fp, vp]),
        "bez_sim_set_flags": (C.c_int, [vp, u32]),
        "bez_sim_set_obs_calls": (C.c_int, [vp, i64]),
        "bez_sim_pre_physics": (C.c_int, [vp, fp, vp]),
        "bez_sim_simulate": (C.c_int, [vp, vp]),
        "bez_sim_post_physics": (C.c_int, [vp, vp]),
        "bez_sim_observe_reward": (C.c_int, [vp, vp]),
        "bez_sim_step": (C.c_int, [vp, fp, vp]),
        "bez_sim_step_many": (C.c_int, [vp, fp, i32, vp]),
        "bez_sim_reset_indexed": (C.c_int, [vp, vp, i32, vp]),
        "bez_sim_set_env_params": (C.c_int, [vp, C.c_int, fp, vp]),
        "bez_sim_get_env_params": (C.c_int, [vp, C.c_int, fp, vp]),
        "bez_sim_set_randomization": (C.c_int, [vp, C.POINTER(abi.BezDrConfig), vp]),
        "bez_sim_add_dr_noise": (C.c_int, [vp, fp, fp, i64, i32, vp]),
        "bez_sim_seed": (C.c_int, [vp, u64]),
        "bez_sim_calibrate": (C.c_int, [vp, u64, i32, vp]),
        "bez_sim_time_steps": (C.c_int, [vp, fp, i32, vp, C.POINTER(C.c_float)]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _LIB = lib
    return lib


EXPORTS = ["bez_sim_default_config", "bez_sim_create", "bez_sim_destroy", "bez_sim_last_error", "bez_sim_get_tensor",
           "bez_sim_refresh_tensor", "bez_sim_set_actor_root_state_tensor_indexed", "bez_sim_set_dof_state_tensor_indexed",
           "bez_sim_set_dof_position_target_tensor", "bez_sim_set_dof_position_target_tensor_indexed",
           "bez_sim_set_net_contact_force_tensor", "bez_sim_set_prev_lin_vel_tensor", "bez_sim_set_goal_tensor", "bez_sim_set_flags",
           "bez_sim_set_obs_calls", "bez_sim_pre_physics", "bez_sim_simulate", "bez_sim_post_physics", "bez_sim_observe_reward", "bez_sim_step",
           "bez_sim_step_many", "bez_sim_reset_indexed", "bez_sim_set_env_params", "bez_sim_get_env_params", "bez_sim_set_randomization", "bez_sim_dr_prelaunch", "bez_sim_dr_step_args", "bez_sim_dr_cancel", "bez_sim_action_noise_source", "bez_sim_add_dr_noise", "bez_sim_seed", "bez_sim_time_steps",
           "bez_sim_calibrate"]
# (the bez_ppo_* entry points of the same library are bound in ppo/fused.py)


class _DevView:
    """__cuda_array_interface__ holder: lets torch wrap a sim-owned device buffer zero-copy
    (the gymtorch.wrap_tensor equivalent, kick_env.py:155-157)."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}
        self._owner = owner  # keeps the sim alive as long as the view lives


class BezSim:
    """One simulator instance on one GPU (one per process / rank)."""

    def __init__(self, cfg: abi.BezSimConfig, device_id: int = 0):
        if not torch.cuda.is_available():
            raise BezSimError("no GPU visible: the bez_kick simulator is HIP-only (sim_device must be a GPU)")
        self.lib = load_library()
        self.cfg = cfg
        self.device_id = int(device_id)
        self.device = torch.device("cuda", self.device_id)
        self.num_envs = int(cfg.num_envs)
        self.has_ball = int(cfg.task) == abi.TASK_KICK
        self.num_actors = 2 if self.has_ball else 1
        self.num_bodies = (29 if int(cfg.flags) & abi.FLAG_CLEATS else 21) + (1 if self.has_ball else 0)
        self.num_obs = 54 if self.has_ball else 52
        h = C.c_void_p()
        rc = self.lib.bez_sim_create(C.byref(cfg), self.device_id, C.byref(h))
        if rc != 0:
            raise BezSimError("bez_sim_create failed (%d): %s" % (rc, self.lib.bez_sim_last_error(None).decode()))
        self.h = h
        self._views = {}

    def close(self):
        if getattr(self, "h", None):
            self._views = {}
            self.lib.bez_sim_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers
    def _check(self, rc):
        if rc != 0:
            raise BezSimError("libbez_sim call failed (%d): %s" % (rc, self.lib.bez_sim_last_error(self.h).decode()))

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _ptr(self, t, dtype, numel=None):
        if t.device != self.device or t.dtype != dtype or not t.is_contiguous():
            raise BezSimError("expected a contiguous %s tensor on %s, got %s on %s" % (dtype, self.device, t.dtype, t.device))
        if numel is not None and t.numel() != numel:
            raise BezSimError("expected %d elements, got %d" % (numel, t.numel()))
        return C.c_void_p(t.data_ptr())

    def tensor(self, which):
        """Zero-copy torch view of a sim-owned buffer (gymtorch.wrap_tensor)."""
        if which not in self._views:
            p = C.c_void_p()
            shape = (C.c_int64 * 3)()
            nd, dt = C.c_int(), C.c_int()
            self._check(self.lib.bez_sim_get_tensor(self.h, which, C.byref(p), shape, C.byref(nd), C.byref(dt)))
            shp = [int(shape[i]) for i in range(nd.value)]
            view = _DevView(p.value, shp, "<f4" if dt.value == abi.DTYPE_F32 else "<i8", self)
            with torch.cuda.device(self.device):
                self._views[which] = torch.as_tensor(view, device=self.device)
        return self._views[which]

    def refresh(self, which):
        self._check(self.lib.bez_sim_refresh_tensor(self.h, which, self._stream()))
        return self.tensor(which)

    # ---- gym.set_* equivalents
    def set_actor_root_state_tensor_indexed(self, root_states, actor_ids):
        self._check(self.lib.bez_sim_set_actor_root_state_tensor_indexed(
            self.h, self._ptr(root_states, torch.float32, self.num_envs * 13 * self.num_actors), self._ptr(actor_ids, torch.int32),
            actor_ids.numel(), self._stream()))

    def set_dof_state_tensor_indexed(self, dof_state, actor_ids):
        self._check(self.lib.bez_sim_set_dof_state_tensor_indexed(
            self.h, self._ptr(dof_state, torch.float32, self.num_envs * 36), self._ptr(actor_ids, torch.int32),
            actor_ids.numel(), self._stream()))

    def set_dof_position_target_tensor(self, targets):
        self._check(self.lib.bez_sim_set_dof_position_target_tensor(
            self.h, self._ptr(targets, torch.float32, self.num_envs * 18), self._stream()))

    def set_dof_position_target_tensor_indexed(self, targets, actor_ids):
        self._check(self.lib.bez_sim_set_dof_position_target_tensor_indexed(
            self.h, self._ptr(targets, torch.float32, self.num_envs * 18), self._ptr(actor_ids, torch.int32),
            actor_ids.numel(), self._stream()))

    def set_net_contact_force_tensor(self, forces):
        self._check(self.lib.bez_sim_set_net_contact_force_tensor(
            self.h, self._ptr(forces, torch.float32, self.num_envs * self.num_bodies * 3), self._stream()))

    def set_prev_lin_vel_tensor(self, prev):
        self._check(self.lib.bez_sim_set_prev_lin_vel_tensor(self.h, self._ptr(prev, torch.float32, self.num_envs * 3), self._stream()))

    def set_goal_tensor(self, goal):
        self._check(self.lib.bez_sim_set_goal_tensor(self.h, self._ptr(goal, torch.float32, self.num_envs * 2), self._stream()))

    def set_flags(self, flags):
        self._check(self.lib.bez_sim_set_flags(self.h, int(flags)))

    def set_obs_calls(self, n):
        self._check(self.lib.bez_sim_set_obs_calls(self.h, int(n)))

    # ---- the path
    def pre_physics(self, actions):
        self._check(self.lib.bez_sim_pre_physics(self.h, self._ptr(actions, torch.float32, self.num_envs * 18), self._stream()))

    def simulate(self):
        self._check(self.lib.bez_sim_simulate(self.h, self._stream()))

    def post_physics(self):
        self._check(self.lib.bez_sim_post_physics(self.h, self._stream()))

    def observe_reward(self):
        self._check(self.lib.bez_sim_observe_reward(self.h, self._stream()))

    def step(self, actions):
        self._check(self.lib.bez_sim_step(self.h, self._ptr(actions, torch.float32, self.num_envs * 18), self._stream()))

    def step_many(self, actions, n_steps):
        self._check(self.lib.bez_sim_step_many(self.h, self._ptr(actions, torch.float32, n_steps * self.num_envs * 18),
                                               n_steps, self._stream()))

    def time_steps(self, actions, n_steps):
        ms = C.c_float()
        self._check(self.lib.bez_sim_time_steps(self.h, self._ptr(actions, torch.float32, n_steps * self.num_envs * 18),
                                                n_steps, self._stream(), C.byref(ms)))
        return float(ms.value)

    def reset_indexed(self, env_ids):
        self._check(self.lib.bez_sim_reset_indexed(self.h, self._ptr(env_ids, torch.int32), env_ids.numel(), self._stream()))

    def set_env_params(self, param, values):
        if values is None:
            self._check(self.lib.bez_sim_set_env_params(self.h, param, None, self._stream()))
        else:
            self._check(self.lib.bez_sim_set_env_params(
                self.h, param, self._ptr(values, torch.float32, self.num_envs * abi.PARAM_WIDTH[param]), self._stream()))

    def get_env_params(self, param):
        """current (N, width) array of a domain-randomisation parameter (defaults where never set)"""
        out = torch.empty(self.num_envs, abi.PARAM_WIDTH[param], device=self.device, dtype=torch.float32)
        self._check(self.lib.bez_sim_get_env_params(self.h, param, C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def set_randomization(self, dr):
        """device-side VecTask.apply_randomizations: `dr` is an abi.BezDrConfig (abi.dr_config_from_params) or None (off)"""
        self._dr_cfg = dr  # keep the struct alive for the call
        self._check(self.lib.bez_sim_set_randomization(self.h, None if dr is None else C.byref(dr), self._stream()))

    def add_dr_noise(self, x, which, out=None):
        """out = x + mean + std * N(0, 1) with the device-resident noise parameters (which: 0 observations, 1 actions); out=None: in place"""
        out = x if out is None else out
        self._check(self.lib.bez_sim_add_dr_noise(self.h, self._ptr(x, torch.float32), self._ptr(out, torch.float32, x.numel()), x.numel(), int(which), self._stream()))
        return out

    def dr_prelaunch(self):
        """The coming step's randomisation kernel now, on torch's current stream (bez_sim_dr_prelaunch): the step then skips its own."""
        self._check(self.lib.bez_sim_dr_prelaunch(self.h, self._stream()))

    def dr_step_args(self):
        """The coming step's randomisation as an opaque argument block (ctypes buffer, BEZ_DR_STEP_BYTES) for a launch of the caller's that
        executes it itself (PolicyForward.rollout_step(dr_step=)); the step then skips its own kernel.  None without a randomisation."""
        blob = (C.c_uint8 * 512)()
        rc = self.lib.bez_sim_dr_step_args(self.h, blob, 512)
        if rc < 0:
            self._check(rc)
        return blob if rc > 0 else None

    def dr_cancel(self):
        """The consumer of dr_prelaunch() / dr_step_args() did not run: the coming step launches its own randomisation kernel again."""
        self._check(self.lib.bez_sim_dr_cancel(self.h))

    def action_noise_source(self):
        """(device pointer of the action-noise snapshot, seed, env id offset) for a consumer that adds the action noise itself, or None
        when the randomisation has no action noise (bez_sim_action_noise_source)"""
        p, seed, off = C.c_void_p(), C.c_uint64(), C.c_int64()
        rc = self.lib.bez_sim_action_noise_source(self.h, C.byref(p), C.byref(seed), C.byref(off))
        if rc < 0:
            self._check(rc)
        return (p.value, seed.value, off.value) if rc == 1 else None

    def seed(self, seed):
        self._check(self.lib.bez_sim_seed(self.h, int(seed)))
