"""ctypes wrapper around the CPU oracle (oracle/bez_oracle.c).  TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the
product package `bez_isaacgym_amd` (which fails loudly without its HIP library instead).
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from bez_isaacgym_amd.abi import BezSimConfig, default_config, NUM_OBS, NUM_DOFS, NUM_BODIES  # noqa: E402


def build(force=False, target=None):
    """Compile libbez_oracle_{f64,f32}.so with gcc (the recipe is oracle/Makefile); target "asan" = the sanitizer build."""
    args = ["make", "-C", _HERE] + (["-B"] if force else []) + ([target] if target else [])
    subprocess.run(args, check=True, stdout=subprocess.DEVNULL)


# BEZ_ORACLE_VARIANT=asan: load libbez_oracle_*_asan.so (gcc -fsanitize=address,undefined; the process needs libasan preloaded,
# tests/test_oracle_sanitizers.py re-runs the golden and known-answer suites that way)
_VARIANT = os.environ.get("BEZ_ORACLE_VARIANT", "")


_LIBS = {}


def _lib(precision):
    if precision not in _LIBS:
        path = os.path.join(_HERE, "libbez_oracle_%s%s.so" % (precision, "_" + _VARIANT if _VARIANT else ""))
        if not os.path.exists(path):
            build(target=_VARIANT or None)
        lib = C.CDLL(path)
        lib.bez_oracle_create.restype = C.c_void_p
        lib.bez_oracle_create.argtypes = [C.POINTER(BezSimConfig)]
        lib.bez_oracle_philox_word.restype = C.c_uint32
        lib.bez_oracle_philox_word.argtypes = [C.c_uint64, C.c_int64, C.c_uint32, C.c_int]
        _LIBS[precision] = lib
    return _LIBS[precision]


def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """N-env CPU oracle.  All getters/setters use the Isaac tensor layouts (fp32 / int64 numpy)."""

    def __init__(self, cfg=None, num_envs=64, precision="f64"):
        self.lib = _lib(precision)
        self.cfg = cfg if cfg is not None else default_config(num_envs)
        self.n = int(self.cfg.num_envs)
        self.h = C.c_void_p(self.lib.bez_oracle_create(C.byref(self.cfg)))
        self.nbe = int(self.lib.bez_oracle_num_bodies(self.h))    # rows per env of the body tensors (22; cleats 30; walk/orient: no ball row)
        self.nobs = int(self.lib.bez_oracle_num_obs(self.h))      # 54 (kick) / 52 (walk, orient)
        self.nact = int(self.lib.bez_oracle_num_actors(self.h))   # 2 (robot + ball) / 1

    def __del__(self):
        try:
            if self.h:
                self.lib.bez_oracle_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- generic helpers
    def _get(self, name, shape, dtype=np.float32):
        out = np.empty(shape, dtype=dtype)
        getattr(self.lib, "bez_oracle_get_" + name)(self.h, _fp(out))
        return out

    def _set(self, name, arr, dtype=np.float32):
        a = np.ascontiguousarray(arr, dtype=dtype)
        getattr(self.lib, "bez_oracle_set_" + name)(self.h, _fp(a))

    root_states = property(lambda s: s._get("root_states", (s.n * s.nact, 13)))
    goal = property(lambda s: s._get("goal", (s.n, 2)))
    dof_state = property(lambda s: s._get("dof_state", (s.n * NUM_DOFS, 2)))
    rigid_body_states = property(lambda s: s._get("rigid_body_states", (s.n * s.nbe, 13)))
    contact_forces = property(lambda s: s._get("contact_forces", (s.n * s.nbe, 3)))
    targets = property(lambda s: s._get("targets", (s.n, NUM_DOFS)))
    prev_lin_vel = property(lambda s: s._get("prev_lin_vel", (s.n, 3)))
    obs = property(lambda s: s._get("obs", (s.n, s.nobs)))
    feet = property(lambda s: s._get("feet", (s.n, 8)))
    rew = property(lambda s: s._get("rew", (s.n,)))
    vlim_margin = property(lambda s: s._get("vlim_margin", (s.n,)))   # rad/s: how close a speed-limit decision of the last step came to its boundary
    reset_buf = property(lambda s: s._get("reset", (s.n,), np.int64))
    progress_buf = property(lambda s: s._get("progress", (s.n,), np.int64))
    timeout_buf = property(lambda s: s._get("timeout", (s.n,), np.int64))

    def set_root_states(self, a): self._set("root_states", a)
    def set_dof_state(self, a): self._set("dof_state", a)
    def set_contact_forces(self, a): self._set("contact_forces", a)
    def set_targets(self, a): self._set("targets", a)
    def set_goal(self, a): self._set("goal", a)
    def set_prev_lin_vel(self, a): self._set("prev_lin_vel", a)
    def set_reset(self, a): self._set("reset", a, np.int64)
    def set_progress(self, a): self._set("progress", a, np.int64)

    randomize_buf = property(lambda s: s._get("randomize", (s.n,), np.int64))
    dr_noise = property(lambda s: s._get("dr_noise", (4,)))

    def set_randomize(self, a): self._set("randomize", a, np.int64)

    def set_randomization(self, dr):
        """BezDrConfig (abi.dr_config_from_params) or None"""
        self.lib.bez_oracle_set_randomization(self.h, None if dr is None else C.byref(dr))

    def get_env_params(self, param):
        from bez_isaacgym_amd.abi import PARAM_WIDTH
        out = np.zeros((self.n, PARAM_WIDTH[param]), np.float32)
        self.lib.bez_oracle_get_env_params(self.h, C.c_int(param), _fp(out))
        return out

    def set_env_params(self, param, values):
        if values is None:
            self.lib.bez_oracle_set_env_params(self.h, C.c_int(param), None)
        else:
            a = np.ascontiguousarray(values, dtype=np.float32)
            self.lib.bez_oracle_set_env_params(self.h, C.c_int(param), _fp(a))

    # ---- the path
    def pre_physics(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32)
        self.lib.bez_oracle_pre_physics(self.h, _fp(a))

    def simulate(self):
        self.lib.bez_oracle_simulate(self.h)

    def post_physics(self):
        self.lib.bez_oracle_post_physics(self.h)

    def observe_reward(self):
        self.lib.bez_oracle_observe_reward(self.h)

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32)
        self.lib.bez_oracle_step(self.h, _fp(a))

    def reset_idx(self, ids):
        a = np.ascontiguousarray(ids, dtype=np.int32)
        self.lib.bez_oracle_reset_idx(self.h, _fp(a), C.c_int(len(a)))

    def set_flags(self, flags):
        self.lib.bez_oracle_set_flags(self.h, C.c_uint32(flags))

    def set_obs_calls(self, n):
        self.lib.bez_oracle_set_obs_calls(self.h, C.c_int64(n))

    def seed(self, s):
        self.lib.bez_oracle_seed(self.h, C.c_uint64(s))

    def forward_dynamics(self, env=0, mode=0, tau=None):
        """Returns (root spatial acc [ang;lin] about torso origin, qdd[18], ball acc [ang;lin], contact forces[22,3])."""
        a0 = np.zeros(6); qdd = np.zeros(NUM_DOFS); ball = np.zeros(6); cf = np.zeros((NUM_BODIES, 3))
        t = None if tau is None else np.ascontiguousarray(tau, dtype=np.float64)
        self.lib.bez_oracle_forward_dynamics(self.h, C.c_int(env), C.c_int(mode), None if t is None else _fp(t),
                                             _fp(a0), _fp(qdd), _fp(ball), _fp(cf))
        return a0, qdd, ball, cf

    def set_env_state_f64(self, env, pos, quat, lin, ang, q, qd):
        s = np.ascontiguousarray(np.concatenate([pos, quat, lin, ang, q, qd]), dtype=np.float64)
        assert s.size == 49
        self.lib.bez_oracle_set_env_state_f64(self.h, C.c_int(env), _fp(s))

    def philox_word(self, seed, genv, episode, k):
        return int(self.lib.bez_oracle_philox_word(C.c_uint64(seed), C.c_int64(genv), C.c_uint32(episode), C.c_int(k)))
