"""Scripted physics scenarios shared by the tests and by tools/getup_probe.py: the reference's three get-up tables played from lying
starts (resources/library/trajectories/trajectories/simulation_getup{front,back,side}.csv via tests/golden/trajectories.json,
`soccer_trajectories.py:56-91`) and the per-DOF limit sweep of `bez_isaacgym/test/test_kick_env.py:142-186`.  Both run through the
SPLIT entry points (pre_physics + simulate) of an Oracle / SimAdapter object, so no fall reset interferes."""
import json
import os

import numpy as np

from bez_isaacgym_amd.utils.trajectories import JOINT_ORDER, Trajectory

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLES = os.path.join(ROOT, "tests", "golden", "trajectories.json")
S = float(np.sqrt(0.5))
# (root quaternion xyzw, start height of the torso origin): the robot is dropped from just above its resting height
STARTS = {"getupfront": ((0.0, S, 0.0, S), 0.10), "getupback": ((0.0, -S, 0.0, S), 0.10), "getupside": ((S, 0.0, 0.0, S), 0.14)}
VARIANTS = {"yaml defaults": {}, "damping 2": {"kd": 2.0}, "effort 5 N*m": {"effort": 5.0},
            "effort 5 N*m, 24.5 rad/s": {"effort": 5.0, "vel_limit": 24.5}}


def up_z(q):
    """z component of the torso's z axis for xyzw quaternions (n, 4)."""
    return 1.0 - 2.0 * (q[:, 0] ** 2 + q[:, 1] ** 2)


def make_backend(backend, cfg, precision="f64"):
    if backend == "oracle":
        from oracle.bez_oracle import Oracle
        return Oracle(cfg, precision=precision)
    from tests.sim_adapter import SimAdapter
    return SimAdapter(cfg)


def lay_down(sim, n, name, rng):
    """Robot lying with every joint at 0 (+- 0.02 rad so that the envs differ), ball parked 2 m to the side."""
    quat, z = STARTS[name]
    rs = sim.root_states.reshape(n, -1, 13).copy()
    rs[:, 0, :] = 0
    rs[:, 0, 2] = z
    rs[:, 0, 3:7] = quat
    if rs.shape[1] > 1:
        rs[:, 1, :] = 0
        rs[:, 1, 0:3] = (0.0, 2.0, 0.08)
        rs[:, 1, 6] = 1.0
    sim.set_root_states(rs.reshape(-1, 13))
    ds = np.zeros((n, 18, 2), np.float32)
    ds[:, :, 0] = rng.uniform(-0.02, 0.02, (n, 18))
    ds[:, 1, 0] = 0
    sim.set_dof_state(ds.reshape(-1, 2))


def play(sim, n, name, model, settle=60, hold=120, trace_env=None):
    ready = dict(zip(JOINT_ORDER, model["dof_default"]))
    tr = Trajectory(json.load(open(TABLES))["simulation_" + name], ready)
    acts = tr.actions(model["dof_default"])
    zero_pose = -np.asarray(model["dof_default"], np.float32)
    seq = [zero_pose] * settle + list(acts) + [np.zeros(18, np.float32)] * hold
    max_z = np.zeros(n); max_up = np.full(n, -1.0)
    rows = []
    for k, a in enumerate(seq):
        sim.pre_physics(np.tile(np.asarray(a, np.float32), (n, 1)))
        sim.simulate()
        if k % 10 == 9 or k == len(seq) - 1:
            rs = sim.root_states.reshape(n, -1, 13)
            z, up = rs[:, 0, 2].astype(np.float64), up_z(rs[:, 0, 3:7].astype(np.float64))
            if k >= settle:
                max_z = np.maximum(max_z, z); max_up = np.maximum(max_up, up)
            if trace_env is not None:
                rows.append((k, float(z[trace_env]), float(up[trace_env])))
    rs = sim.root_states.reshape(n, -1, 13)
    z, up = rs[:, 0, 2].astype(np.float64), up_z(rs[:, 0, 3:7].astype(np.float64))
    ok = np.isfinite(z) & np.isfinite(up)
    return dict(steps=len(seq), final_z=float(np.median(z)), final_up=float(np.median(up)), max_z=float(np.median(max_z)),
                max_up=float(np.median(max_up)), standing=float(((z > 0.28) & (up > 0.9) & ok).mean()), finite=float(ok.mean()),
                final_z_minmax=[float(z.min()), float(z.max())], trace=rows)




def dof_sweep(sim, n, model, speed=3.0, hold=25, dofs=range(18)):
    """`test_motor_action_agent` (test_kick_env.py:142-186): one DOF after the other, the commanded position moves at `speed` rad/s
    to the lower limit, to the upper limit and back to the default pose while the others hold theirs.  The reference runs it on the
    floating robot and notes "better when fixBaseLink = True"; here the robot floats in zero gravity one metre above the plane (set
    cfg.gravity = 0 before creating `sim`), which takes the ground out of it without pinning the base.
    Returns per DOF: closest approach to the lower / upper limit and the final distance to the default pose (max over envs)."""
    lo, hi, dflt = (np.asarray(model[k], np.float64) for k in ("dof_lower", "dof_upper", "dof_default"))
    dt = float(sim.cfg.dt)
    out = []

    def place():
        rs = sim.root_states.reshape(n, -1, 13).copy()
        rs[:, 0, :] = 0; rs[:, 0, 2] = 1.0; rs[:, 0, 6] = 1.0
        if rs.shape[1] > 1:
            rs[:, 1, :] = 0; rs[:, 1, 0:3] = (0.0, 3.0, 0.08); rs[:, 1, 6] = 1.0
        sim.set_root_states(rs.reshape(-1, 13))
        ds = np.zeros((n, 18, 2), np.float32); ds[:, :, 0] = dflt
        sim.set_dof_state(ds.reshape(-1, 2))

    def run(cmd, d, goal):
        best = np.full(n, np.inf)
        steps = int(abs(goal - cmd[d]) / (speed * dt)) + 1
        for k in range(steps + hold):
            cmd[d] = goal if k >= steps else cmd[d] + np.sign(goal - cmd[d]) * min(speed * dt, abs(goal - cmd[d]))
            sim.pre_physics(np.tile((cmd - dflt).astype(np.float32), (n, 1)))
            sim.simulate()
            if k >= steps:
                q = sim.dof_state.reshape(n, 18, 2)[:, d, 0].astype(np.float64)
                best = np.minimum(best, np.abs(q - goal))
        return best
    for d in dofs:
        place()  # every DOF starts from the default pose at rest (a blocked sweep may leave the floating robot drifting)
        cmd = dflt.copy()
        miss_lo = run(cmd, d, lo[d])
        miss_hi = run(cmd, d, hi[d])
        back = run(cmd, d, dflt[d])
        q = sim.dof_state.reshape(n, 18, 2)[:, :, 0]
        out.append(dict(dof=int(d), name=JOINT_ORDER[d], miss_lower=float(miss_lo.max()), miss_upper=float(miss_hi.max()),
                        miss_default=float(back.max()), finite=bool(np.isfinite(q).all())))
    return out
