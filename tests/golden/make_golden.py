#!/usr/bin/env python3
"""Generates tests/golden/kick_env_golden.npz by RUNNING THE REFERENCE'S OWN CODE.

Build-container only (needs /root/reference; the GPU box never runs this).  The reference's
`tasks.kick_env` imports the closed `isaacgym` package and `gym`; neither is installed, so a
throw-away stub package is written to a temp dir (this file contains the stub's text -- it is this
build's code, not the reference's).  What the stub supplies and the reference then star-imports
through utils/torch_jit_utils.py:31 are the `isaacgym.torch_utils` helpers [ext]:
quat_rotate / get_basis_vector / get_euler_xyz / normalize_angle / tensor_clamp / torch_rand_float /
to_torch / get_axis_params / quat_conjugate ...  Their formulas are the one un-pinned piece (Isaac Gym
is not under /root/reference), so the fixture also records get_euler_xyz / quat_rotate outputs.

Everything else recorded here is computed by reference code, unmodified, imported from
/root/reference/bez_isaacgym:
  * the TorchScript functions compute_imu, compute_off_orn, compute_feet_sensors_no_cleats,
    compute_bez_observations, compute_bez_reward (kick_env.py:857-1417) on seeded synthetic tensors;
  * KickEnv.pre_physics_step (kick_env.py:410-419) and the whole of VecTask.step
    (vec_task.py:303-349 -> kick_env.py:426-438) driven on a KickEnv object whose simulator is a
    scripted fake (gym.simulate writes the next pre-generated sim state into the wrapped tensors),
    which pins action clamp, head zeroing, target clamp, timeout/progress bookkeeping, the obs
    assembly, reward/reset ordering and the prev_lin_vel aliasing (quirk Q1) over multi-step
    sequences.

Only inputs and outputs (numbers) are saved.
"""
import os
import sys
import tempfile
import types

import numpy as np

REF = os.environ.get("BEZ_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kick_env_golden.npz")

STUB_TORCH_UTILS = r'''
import numpy as np
import torch
from torch import Tensor
from typing import Tuple, List

def to_torch(x, dtype=torch.float, device='cpu', requires_grad=False):
    return torch.tensor(x, dtype=dtype, device=device, requires_grad=requires_grad)

@torch.jit.script
def quat_mul(a, b):
    shape = a.shape
    a = a.reshape(-1, 4); b = b.reshape(-1, 4)
    x1, y1, z1, w1 = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    x2, y2, z2, w2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    ww = (z1 + x1) * (x2 + y2); yy = (w1 - y1) * (w2 + z2); zz = (w1 + y1) * (w2 - z2)
    xx = ww + yy + zz
    qq = 0.5 * (xx + (z1 - x1) * (x2 - y2))
    w = qq - ww + (z1 - y1) * (y2 - z2); x = qq - xx + (x1 + w1) * (x2 + w2)
    y = qq - yy + (w1 - x1) * (y2 + z2); z = qq - zz + (z1 + y1) * (w2 - x2)
    return torch.stack([x, y, z, w], dim=-1).view(shape)

@torch.jit.script
def normalize(x, eps: float = 1e-9):
    return x / x.norm(p=2, dim=-1).clamp(min=eps, max=None).unsqueeze(-1)

@torch.jit.script
def quat_conjugate(a):
    shape = a.shape
    a = a.reshape(-1, 4)
    return torch.cat((-a[:, :3], a[:, -1:]), dim=-1).view(shape)

@torch.jit.script
def quat_rotate(q, v):
    shape = q.shape
    q_w = q[:, -1]
    q_vec = q[:, :3]
    a = v * (2.0 * q_w ** 2 - 1.0).unsqueeze(-1)
    b = torch.cross(q_vec, v, dim=-1) * q_w.unsqueeze(-1) * 2.0
    c = q_vec * torch.bmm(q_vec.view(shape[0], 1, 3), v.view(shape[0], 3, 1)).squeeze(-1) * 2.0
    return a + b + c

@torch.jit.script
def quat_rotate_inverse(q, v):
    shape = q.shape
    q_w = q[:, -1]
    q_vec = q[:, :3]
    a = v * (2.0 * q_w ** 2 - 1.0).unsqueeze(-1)
    b = torch.cross(q_vec, v, dim=-1) * q_w.unsqueeze(-1) * 2.0
    c = q_vec * torch.bmm(q_vec.view(shape[0], 1, 3), v.view(shape[0], 3, 1)).squeeze(-1) * 2.0
    return a - b + c

@torch.jit.script
def copysign(a, b):
    # type: (float, Tensor) -> Tensor
    a = torch.tensor(a, device=b.device, dtype=torch.float).repeat(b.shape[0])
    return torch.abs(a) * torch.sign(b)

@torch.jit.script
def get_euler_xyz(q):
    qx, qy, qz, qw = 0, 1, 2, 3
    sinr_cosp = 2.0 * (q[:, qw] * q[:, qx] + q[:, qy] * q[:, qz])
    cosr_cosp = q[:, qw] * q[:, qw] - q[:, qx] * q[:, qx] - q[:, qy] * q[:, qy] + q[:, qz] * q[:, qz]
    roll = torch.atan2(sinr_cosp, cosr_cosp)
    sinp = 2.0 * (q[:, qw] * q[:, qy] - q[:, qz] * q[:, qx])
    pitch = torch.where(torch.abs(sinp) >= 1, copysign(np.pi / 2.0, sinp), torch.asin(sinp))
    siny_cosp = 2.0 * (q[:, qw] * q[:, qz] + q[:, qx] * q[:, qy])
    cosy_cosp = q[:, qw] * q[:, qw] + q[:, qx] * q[:, qx] - q[:, qy] * q[:, qy] - q[:, qz] * q[:, qz]
    yaw = torch.atan2(siny_cosp, cosy_cosp)
    return roll % (2 * np.pi), pitch % (2 * np.pi), yaw % (2 * np.pi)

@torch.jit.script
def normalize_angle(x):
    return torch.atan2(torch.sin(x), torch.cos(x))

@torch.jit.script
def tensor_clamp(t, min_t, max_t):
    return torch.max(torch.min(t, max_t), min_t)

@torch.jit.script
def get_basis_vector(q, v):
    return quat_rotate(q, v)

def get_axis_params(value, axis_idx, x_value=0., dtype=float, n_dims=3):
    zs = np.zeros((n_dims,))
    params = np.where(zs == 1., value, zs)
    params[axis_idx] = value
    params[0] = x_value if axis_idx != 0 else params[0]
    return list(params.astype(dtype))

@torch.jit.script
def torch_rand_float(lower, upper, shape, device):
    # type: (float, float, Tuple[int, int], str) -> Tensor
    return (upper - lower) * torch.rand(*shape, device=device) + lower
'''

STUB_GYMAPI = r'''
class _Any:
    def __init__(self, *a, **k): pass
    def __getattr__(self, n): return _Any()
    def __call__(self, *a, **k): return _Any()
class SimParams(_Any): pass
def __getattr__(name): return _Any
'''
STUB_GYMUTIL = "\n".join("def %s(*a, **k): return None" % n for n in (
    "get_property_setter_map", "get_property_getter_map", "get_default_setter_args", "apply_random_samples",
    "check_buckets", "generate_random_samples"))
STUB_GYMTORCH = "def wrap_tensor(t): return t\ndef unwrap_tensor(t): return t\n"
STUB_SPACES = ("class Box:\n    def __init__(self, low, high):\n        import numpy as np\n"
               "        self.low = np.asarray(low); self.high = np.asarray(high); self.shape = self.low.shape\n")


def install_stub():
    d = tempfile.mkdtemp(prefix="bez_golden_stub_")
    os.makedirs(os.path.join(d, "isaacgym"))
    os.makedirs(os.path.join(d, "gym"))
    files = {"isaacgym/__init__.py": "", "isaacgym/gymtorch.py": STUB_GYMTORCH, "isaacgym/gymapi.py": STUB_GYMAPI,
             "isaacgym/gymutil.py": STUB_GYMUTIL, "isaacgym/torch_utils.py": STUB_TORCH_UTILS,
             "gym/__init__.py": "from . import spaces\nclass Space: pass\n", "gym/spaces.py": STUB_SPACES}
    for k, v in files.items():
        with open(os.path.join(d, k), "w") as f:
            f.write(v)
    sys.path.insert(0, d)
    if not hasattr(np, "Inf"):
        np.Inf = np.inf
    import matplotlib
    matplotlib.use = lambda *a, **k: None
    sys.path.insert(0, os.path.join(REF, "bez_isaacgym"))
    return d


def rand_quat(rng, n, tilt=1.0):
    q = rng.normal(size=(n, 4)).astype(np.float32)
    q[:, :2] *= tilt
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q.astype(np.float32)


def main():
    import json
    install_stub()
    import torch
    torch.manual_seed(0)
    import tasks.kick_env as K
    from tasks.base.vec_task import VecTask
    model = json.load(open(os.path.join(os.path.dirname(OUT), "..", "..", "bez_isaacgym_amd", "model", "bez_model.json")))
    rng = np.random.default_rng(20261003)
    T = torch.from_numpy
    G = {}
    N = 64
    dt = 0.01667
    default = np.tile(np.array(model["dof_default"], np.float32), (N, 1))
    lower = np.array(model["dof_lower"], np.float32)
    upper = np.array(model["dof_upper"], np.float32)
    goal = np.tile(np.array([[1.5, 0.0]], np.float32), (N, 1))
    ball_init = np.tile(np.array([[0.175, 0.0]], np.float32), (N, 1))
    bez_init_xy = np.array([0.0, 0.0], np.float32)
    gravity_vec = np.tile(np.array([[0, 0, -1.0]], np.float32), (N, 1))
    up_vec = np.tile(np.array([[0, 0, 1.0]], np.float32), (N, 1))
    inv_start_rot = np.tile(np.array([[0, 0, 0, 1.0]], np.float32), (N, 1))

    # ---------------- helper pins [ext]
    q = rand_quat(rng, N)
    r, p, y = K.get_euler_xyz(T(q))
    G["ext_quat"] = q
    G["ext_euler"] = np.stack([r.numpy(), p.numpy(), y.numpy()], 1)
    v = rng.normal(size=(N, 3)).astype(np.float32)
    G["ext_vec"] = v
    G["ext_quat_rotate"] = K.quat_rotate(T(q), T(v)).numpy()

    # ---------------- compute_imu (kick_env.py:888-930)
    q = rand_quat(rng, N)
    vel = (rng.normal(size=(N, 3)) * 0.5).astype(np.float32)
    ang = (rng.normal(size=(N, 3)) * 4.0).astype(np.float32)
    ang[:4] *= 5.0  # exercise the +-8.7266 clamp
    prev = vel + (rng.normal(size=(N, 3)) * 0.1).astype(np.float32)
    prev[:8] = vel[:8] + rng.normal(size=(8, 3)).astype(np.float32)  # exercise the +-19.62 clamp
    imu, newprev = K.compute_imu(T(q), T(vel), T(ang), T(prev), T(gravity_vec), T(inv_start_rot), 2. * 9.81, 8.7266, dt, N)
    G.update(imu_quat=q, imu_vel=vel, imu_ang=ang, imu_prev=prev, imu_out=imu.numpy(), imu_newprev=newprev.numpy())
    # first call of a process: prev is an int64 zeros tensor (kick_env.py:183)
    prev_i64 = torch.tensor([[0, 0, 0]]).repeat((N, 1))
    imu0, _ = K.compute_imu(T(q), T(vel), T(ang), prev_i64, T(gravity_vec), T(inv_start_rot), 2. * 9.81, 8.7266, dt, N)
    G["imu_out_first"] = imu0.numpy()
    # aliased prev (what every later step sees, quirk Q1)
    tv = T(vel)
    imu_alias, _ = K.compute_imu(T(q), tv, T(ang), tv, T(gravity_vec), T(inv_start_rot), 2. * 9.81, 8.7266, dt, N)
    G["imu_out_alias"] = imu_alias.numpy()

    # ---------------- compute_off_orn (kick_env.py:933-962)
    pos = (rng.normal(size=(N, 3)) * 0.3).astype(np.float32)
    pos[:, 2] = 0.3 + 0.05 * pos[:, 2]
    q = rand_quat(rng, N, tilt=0.3)
    G.update(orn_pos=pos, orn_quat=q, orn_out=K.compute_off_orn(T(pos), T(q), T(goal)).numpy())

    # ---------------- compute_feet_sensors_no_cleats (kick_env.py:966-1040)
    vals = np.array([0.0, 0.005, -0.005, 0.01, 0.0100001, -0.02, 0.02, 0.5, 0.99, 1.0, 1.0001, 1.5, -1.5, 30.0], np.float32)
    forces = vals[rng.integers(0, len(vals), size=(N, 3))].astype(np.float32)
    forces[:14, 2] = vals  # sweep fz over every threshold value
    f_in = forces.copy()
    ft = T(forces)  # mutated in place by the reference (kick_env.py:987-990)
    dev = "cpu"
    cases = [[1., -1., -1., -1.], [-1., -1., 1., -1.], [1., -1., 1., -1.], [-1., 1., -1., -1.], [-1., -1., -1., 1.],
             [-1., 1., -1., 1.], [1., 1., -1., -1.], [-1., -1., 1., 1.], [1., 1., 1., 1.], [-1., -1., -1., -1.]]
    out = K.compute_feet_sensors_no_cleats(ft, torch.tensor([[-1.] * 4]).repeat((N, 1)), torch.ones(1), torch.zeros(1),
                                           torch.zeros(3), *[torch.tensor(c) for c in cases])
    G.update(feet_in=f_in, feet_out=out.numpy(), feet_filtered=ft.numpy().copy())

    # ---------------- compute_bez_reward (kick_env.py:1198-1395) + observations
    def reward_case(tag, n_mod):
        dof_pos = (default + rng.normal(size=(N, 18)).astype(np.float32) * 0.3).astype(np.float32)
        dof_vel = rng.normal(size=(N, 18)).astype(np.float32)
        v_imu = (rng.normal(size=(N, 3)) * 0.4).astype(np.float32)
        w_imu = rng.normal(size=(N, 3)).astype(np.float32)
        root = np.zeros((N, 3), np.float32)
        root[:, :2] = rng.normal(size=(N, 2)) * 0.15
        root[:, 2] = 0.32 + rng.normal(size=N) * 0.02
        qi = rand_quat(rng, N, tilt=0.2)
        ball = np.zeros((N, 3), np.float32)
        ball[:, 0] = 0.175 + np.abs(rng.normal(size=N)) * 0.3
        ball[:, 1] = rng.normal(size=N) * 0.2
        ball[:, 2] = 0.08
        ball_v = (rng.normal(size=(N, 3)) * 0.8).astype(np.float32)
        reset = np.zeros(N, np.int64)
        progress = rng.integers(1, 890, size=N).astype(np.int64)
        feet = rng.choice(np.array([-1.0, 1.0], np.float32), size=(N, 8)).astype(np.float32)
        n_mod(dict(root=root, ball=ball, progress=progress, reset=reset, ball_v=ball_v))
        rew, rst = K.compute_bez_reward(T(dof_pos), T(dof_vel), T(default), T(v_imu), T(w_imu), T(root), T(qi), T(up_vec),
                                        T(ball), T(ball_v), T(goal), T(ball_init), T(bez_init_xy), T(reset), T(progress),
                                        T(feet), 900, N)
        for k, a in dict(dof_pos=dof_pos, dof_vel=dof_vel, v_imu=v_imu, w_imu=w_imu, root=root, quat=qi, ball=ball,
                         ball_v=ball_v, reset=reset, progress=progress, feet=feet, rew=rew.numpy(), rst=rst.numpy()).items():
            G["rew_%s_%s" % (tag, k)] = a

    reward_case("normal", lambda d: None)

    def edge(d):
        d["root"][0:6, 2] = [0.274, 0.2749, 0.2751, 0.276, 0.20, 0.10]          # fall threshold 0.275
        d["root"][6:10, 0] = [0.49, 0.51, -0.6, 0.0]; d["root"][6:10, 1] = [0.0, 0.0, 0.0, 0.52]  # out of bound 0.5
        d["ball"][10:14, 0] = [1.6, 1.52, 1.5, 1.46]; d["ball"][10:14, 1] = [1.0, 0.5, 0.04, 0.0]  # angle diff / goal
        d["ball"][14:18, 0] = [1.5, 1.47, 1.53, 1.5]; d["ball"][14:18, 1] = [0.049, 0.0, 0.03, -0.051]  # goal radius 0.05
        d["progress"][14:18] = [100, 450, 899, 10]
        d["progress"][18:22] = [899, 900, 901, 1000]                              # horizon
        d["reset"][22:26] = 1                                                      # incoming reset flags stay
        d["ball"][26:30, 0] = [0.175, 0.47, 0.48, 0.30]; d["ball"][26:30, 1] = [0.0, 0.0, 0.05, 0.27]  # kicked 0.3
        d["root"][30, 2] = 0.1; d["ball"][30, 0] = 1.5; d["ball"][30, 1] = 0.0; d["progress"][30] = 950  # all at once
    reward_case("edge", edge)

    # compute_bez_observations (kick_env.py:1398-1417)
    parts = [rng.normal(size=(N, k)).astype(np.float32) for k in (18, 18, 6, 2, 8)]
    obs = K.compute_bez_observations(*[T(p_) for p_ in parts], T(ball_init))
    G["obscat_in"] = np.concatenate(parts, 1)
    G["obscat_out"] = obs.numpy()

    # ---------------- KickEnv.pre_physics_step + VecTask.step on a scripted fake simulator
    class FakeGym:
        def __init__(self):
            self.targets = None
            self.script = None
            self.t = 0
            self.env = None
        def set_dof_position_target_tensor(self, sim, t):
            self.targets = t.clone()
        def simulate(self, sim):
            s = self.script[self.t]
            e = self.env
            e.root_states[:] = T(s["root"]); e.dof_state[:] = T(s["dof"]); e.rigid_body[:] = T(s["rb"]); e.contact[:] = T(s["cf"])
            self.t += 1
        def fetch_results(self, *a): pass
        def __getattr__(self, name):  # refresh_* etc.
            return lambda *a, **k: None

    def make_env(n):
        e = object.__new__(K.KickEnv)
        e.cfg = {"env": {}, "sim": {}}
        e.device = "cpu"; e.rl_device = "cpu"; e.num_environments = n
        e.num_observations = 54; e.num_actions = 18; e.num_states = 0; e.control_freq_inv = 1
        e.clip_obs = np.inf; e.clip_actions = 3.9
        e.viewer = None; e.dr_randomizations = {}; e.randomize = False; e.debug_rewards = False; e.cleats = False
        e.gym = FakeGym(); e.gym.env = e; e.sim = None
        VecTask.allocate_buffers(e)
        e.obs_dict = {}
        e.dt = dt; e.max_episode_length = 900; e.num_dof = 18
        e.imu_max_ang_vel = 8.7266; e.imu_max_lin_acc = 2. * 9.81
        e.root_states = torch.zeros(n * 2, 13); e.dof_state = torch.zeros(n * 18, 2)
        e.rigid_body = torch.zeros(n * 22, 13); e.contact = torch.zeros(n * 22, 3)
        e.goal = torch.tensor([[1.5, 0.0]]).repeat((n, 1)); e.bez_init_xy = torch.tensor([0.0, 0.0])
        e.ball_init = torch.tensor([[0.175, 0.0]]).repeat((n, 1))
        # the views of kick_env.py:168-196
        e.dof_pos_bez = e.dof_state.view(n, 18, -1)[..., 0]; e.dof_vel_bez = e.dof_state.view(n, 18, -1)[..., 1]
        e.root_pos_bez = e.root_states.view(n, -1, 13)[..., 0, 0:3]
        e.root_orient_bez = e.rigid_body.view(n, -1, 13)[..., 1, 3:7]
        e.root_vel_bez = e.rigid_body.view(n, -1, 13)[..., 1, 7:10]
        e.root_ang_bez = e.rigid_body.view(n, -1, 13)[..., 1, 10:13]
        e.root_pos_ball = e.root_states.view(n, -1, 13)[..., 1, 0:3]
        e.root_orient_ball = e.root_states.view(n, -1, 13)[..., 1, 3:7]
        e.root_vel_ball = e.root_states.view(n, -1, 13)[..., 1, 7:10]
        e.prev_lin_vel = torch.tensor([[0, 0, 0]]).repeat((n, 1))
        e.feet = torch.tensor([[-1.] * 8]).repeat((n, 1))
        e.left_foot_contact_forces = e.contact.view(n, -1, 3)[..., 12, 0:3]
        e.right_foot_contact_forces = e.contact.view(n, -1, 3)[..., 20, 0:3]
        e.default_dof_pos = T(default[:n].copy())
        e.dof_pos_limits_lower = T(lower.copy()); e.dof_pos_limits_upper = T(upper.copy())
        e.actions = torch.zeros(n, 18)
        e.gravity_vec = T(gravity_vec[:n].copy()); e.up_vec = T(up_vec[:n].copy()); e.inv_start_rot = T(inv_start_rot[:n].copy())
        e.reset_buf[:] = 0  # state after KickEnv.__init__'s reset_idx(all) (kick_env.py:238,850)
        return e

    # pre_physics_step alone
    e = make_env(N)
    acts = (rng.uniform(-5, 5, size=(N, 18))).astype(np.float32)
    K.KickEnv.pre_physics_step(e, torch.clamp(T(acts), -3.9, 3.9))
    G["pre_actions"] = acts
    G["pre_targets"] = e.gym.targets.numpy()

    # scripted multi-step sequence through VecTask.step (no resets triggered: states stay healthy)
    S = 6
    script = []
    for t in range(S):
        root = np.zeros((N, 2, 13), np.float32)
        root[:, 0, 0:2] = rng.normal(size=(N, 2)) * 0.05; root[:, 0, 2] = 0.32 + rng.normal(size=N) * 0.01
        root[:, 0, 3:7] = rand_quat(rng, N, tilt=0.1)
        root[:, 0, 7:10] = rng.normal(size=(N, 3)) * 0.2; root[:, 0, 10:13] = rng.normal(size=(N, 3))
        root[:, 1, 0] = 0.175 + 0.02 * t + np.abs(rng.normal(size=N)) * 0.02; root[:, 1, 1] = rng.normal(size=N) * 0.02
        root[:, 1, 2] = 0.08; root[:, 1, 6] = 1.0
        root[:, 1, 7:10] = rng.normal(size=(N, 3)) * 0.3
        dof = np.zeros((N, 18, 2), np.float32)
        dof[:, :, 0] = default + rng.normal(size=(N, 18)) * 0.1; dof[:, :, 1] = rng.normal(size=(N, 18))
        rb = (rng.normal(size=(N, 22, 13))).astype(np.float32)
        rb[:, 1, :] = root[:, 0, :]  # IMU link rides on the torso origin (soccerbot_stl.urdf:567-572)
        rb[:, 0, :] = root[:, 0, :]
        cf = (rng.normal(size=(N, 22, 3)) * 0.02).astype(np.float32)
        cf[:, 12, 2] = np.abs(rng.normal(size=N)) * 10; cf[:, 20, 2] = np.abs(rng.normal(size=N)) * 10
        cf[: N // 4, 12, :] = 0.0
        script.append(dict(root=root.reshape(N * 2, 13), dof=dof.reshape(N * 18, 2), rb=rb.reshape(N * 22, 13),
                           cf=cf.reshape(N * 22, 3)))
    e = make_env(N)
    e.gym.script = script
    e.progress_buf[:] = T(rng.integers(0, 880, size=N).astype(np.int64))
    e.progress_buf[0:3] = torch.tensor([897, 896, 895])  # time-out bookkeeping within the sequence
    G["seq_progress0"] = e.progress_buf.numpy().copy()
    seq_actions = rng.uniform(-4.5, 4.5, size=(S, N, 18)).astype(np.float32)
    for t in range(S):
        obs_dict, rew, rst, extras = VecTask.step(e, T(seq_actions[t]))
        G["seq%d_root" % t] = script[t]["root"]; G["seq%d_dof" % t] = script[t]["dof"]; G["seq%d_cf" % t] = script[t]["cf"]
        G["seq%d_targets" % t] = e.gym.targets.numpy().copy()
        G["seq%d_obs" % t] = obs_dict["obs"].numpy().copy(); G["seq%d_rew" % t] = rew.numpy().copy()
        G["seq%d_reset" % t] = rst.numpy().copy(); G["seq%d_timeout" % t] = extras["time_outs"].numpy().copy()
        G["seq%d_progress" % t] = e.progress_buf.numpy().copy()
        G["seq%d_cf_after" % t] = e.contact.numpy().copy()
        # keep the sequence reset-free: the reference's reset_idx draws from torch's global RNG, which
        # no counter-based generator can reproduce
        e.reset_buf[:] = 0
    G["seq_actions"] = seq_actions
    G["seq_len"] = np.array(S)

    np.savez_compressed(OUT, **G)
    print("wrote", OUT, "with", len(G), "arrays,", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
