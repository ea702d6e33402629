"""CPU: known-answer and invariant tests of the oracle's physics (the reference pins none: SURVEY.md section 4).

  * bare ABA vs an independent numpy RNEA in link-local coordinates (tests/rbd_numpy.py)
  * conservation laws of the contact-free, drive-free model integrated with RK4
  * Philox4x32-10 known answers (Random123 kat vectors)
  * behaviour of the full model: standing, ball drop / rest / roll, PD step response, weight on the feet
"""
import numpy as np
import pytest

from bez_isaacgym_amd import abi
from oracle.bez_oracle import Oracle
from tests import rbd_numpy as R

G = np.array([0.0, 0.0, float(np.float32(-9.81))])  # BezSimConfig.gravity is fp32


def _rand_state(rng, scale_v=1.0):
    quat = rng.normal(size=4); quat /= np.linalg.norm(quat)
    v0 = np.concatenate([rng.normal(size=3) * 2.0, rng.normal(size=3)]) * scale_v
    q = rng.uniform(-1.0, 1.0, 18)
    qd = rng.normal(size=18) * 2.0 * scale_v
    return quat, v0, q, qd


def _inject(o, pos, quat, v0, q, qd):
    """Full-precision state injection (v0 = [w; v]); the ball is parked far away."""
    quat = np.asarray(quat, float) / np.linalg.norm(quat)
    o.set_env_state_f64(0, pos, quat, v0[3:], v0[:3], q, qd)
    return np.asarray(pos, float), quat, np.asarray(v0, float), np.asarray(q, float), np.asarray(qd, float)


def test_aba_matches_independent_rnea(model):
    """Forward dynamics (oracle, world-aligned ABA) plugged into inverse dynamics (numpy RNEA, link-local):
    must return the applied joint torques and a zero base wrench."""
    rng = np.random.default_rng(0)
    o = Oracle(num_envs=1)
    for trial in range(20):
        quat, v0, q, qd = _rand_state(rng)
        tau = rng.normal(size=18) * 0.5
        pos, quat, v0, q, qd = _inject(o, [0.1, -0.2, 0.6], quat, v0, q, qd)
        a0, qdd, _, _ = o.forward_dynamics(0, 1, tau)
        f0, tau_back = R.rnea_floating(model, quat, v0, a0, q, qd, qdd, G)
        np.testing.assert_allclose(tau_back, tau, atol=1e-7)  # 7000:1 mass ratios: fp64 round-off ~1e-8
        np.testing.assert_allclose(f0, 0.0, atol=1e-7)


def test_free_flight_conservation(model):
    """No drives, no contact: RK4 on the oracle's accelerations conserves total energy and angular momentum about
    the COM; linear momentum changes by exactly m g t."""
    rng = np.random.default_rng(1)
    o = Oracle(num_envs=1)
    quat, v0, q, qd = _rand_state(rng, scale_v=0.5)
    pos = np.array([0.0, 0.0, 2.0])

    def deriv(s):
        p, qu, v, qq, qqd = s[0:3], s[3:7], s[7:13], s[13:31], s[31:49]
        p_, qu_, v_, qq_, qqd_ = _inject(o, p, qu / np.linalg.norm(qu), v, qq, qqd)
        a0, qdd, _, _ = o.forward_dynamics(0, 1, np.zeros(18))
        w, vl = v[:3], v[3:]
        x, y, z, s_ = qu
        dq = 0.5 * np.array([w[0] * s_ + w[1] * z - w[2] * y, -w[0] * z + w[1] * s_ + w[2] * x,
                             w[0] * y - w[1] * x + w[2] * s_, -w[0] * x - w[1] * y - w[2] * z])
        return np.concatenate([vl, dq, a0[:3], a0[3:] + np.cross(w, vl), qqd, qdd])

    s = np.concatenate([pos, quat, v0, q, qd])
    h, steps = 5e-4, 400

    def mech(s):
        return R.mechanics(model, s[0:3], s[3:7] / np.linalg.norm(s[3:7]), s[7:13], s[13:31], s[31:49], G)
    m0 = mech(s)
    for _ in range(steps):
        k1 = deriv(s); k2 = deriv(s + 0.5 * h * k1); k3 = deriv(s + 0.5 * h * k2); k4 = deriv(s + h * k3)
        s = s + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
    m1 = mech(s)
    T = h * steps
    E0, E1 = m0["KE"] + m0["PE"], m1["KE"] + m1["PE"]
    assert abs(E1 - E0) < 1e-8 * max(1.0, abs(E0)), (E0, E1)
    np.testing.assert_allclose(m1["P"], m0["P"] + m0["mass"] * G * T, atol=1e-9)
    Lc0 = m0["L"] - np.cross(m0["com"], m0["P"]); Lc1 = m1["L"] - np.cross(m1["com"], m1["P"])
    np.testing.assert_allclose(Lc1, Lc0, atol=1e-9)
    assert abs(m0["mass"] - model["total_mass"]) < 1e-12


def test_philox_known_answers():
    """Random123 kat_vectors, philox4x32-10."""
    o = Oracle(num_envs=1)
    assert [o.philox_word(0, 0, 0, k) for k in range(4)] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    ones = [o.philox_word(0xffffffffffffffff, -1, 0xffffffff, -4 + k) for k in range(4)]
    assert ones == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]


def test_reset_distribution_and_sharding():
    """kick_env.py:786-791: q = clamp(default + U(-.15,.15)), qd = U(-.1,.1); keyed by global env id."""
    a = Oracle(abi.default_config(512, seed=9))
    b = Oracle(abi.default_config(256, seed=9, env_id_offset=256))
    da, db = a.dof_state.reshape(512, 18, 2), b.dof_state.reshape(256, 18, 2)
    np.testing.assert_array_equal(da[256:], db)
    dflt = np.array([0, 0, 0, 1.5, 0, 0, .564, -1.176, .613, 0, 0, 1.5, 0, 0, .564, -1.176, .613, 0], np.float32)
    off = da[:, :, 0] - dflt
    assert off.min() >= -0.1500001 and off.max() <= 0.1500001 and abs(off.mean()) < 0.01 and off.std() > 0.07
    assert np.abs(da[:, :, 1]).max() <= 0.1000001 and da[:, :, 1].std() > 0.05
    c = Oracle(abi.default_config(512, seed=10))
    assert not np.array_equal(c.dof_state, a.dof_state)
    ids = np.array([3, 5], np.int32)
    before = a.dof_state.reshape(512, 18, 2).copy()
    a.reset_idx(ids)
    after = a.dof_state.reshape(512, 18, 2)
    assert not np.array_equal(before[3], after[3]) and np.array_equal(before[4], after[4])  # new episode -> new draw


def test_standing_full_episode():
    """Scenario of the reference's zero-action smoke test (test/test_kick_env.py:96-112), with assertions: the
    ready pose stands for the whole 900-step episode; the feet carry the weight; the horizon reset fires."""
    o = Oracle(num_envs=4)
    act = np.zeros((4, 18), np.float32)
    for t in range(899):
        o.step(act)
        assert (o.reset_buf == 0).all(), t
    rs = o.root_states.reshape(4, 2, 13)
    assert np.all(np.abs(rs[:, 0, 2] - 0.3235) < 0.01) and np.all(np.linalg.norm(rs[:, 0, :2], axis=1) < 0.05)
    assert np.all(np.abs(rs[:, 0, 7:13]) < 1e-3)
    cf = o.contact_forces.reshape(4, 22, 3)
    np.testing.assert_allclose(cf[:, 12, 2] + cf[:, 20, 2], 2.827994 * 9.81, rtol=5e-3)
    assert np.all(np.abs(cf[:, [12, 20], :2]) < 0.5)
    np.testing.assert_array_equal(o.obs[:, 44:52], np.ones((4, 8), np.float32))  # fx=fy=0 after the noise gate -> case 11
    np.testing.assert_allclose(rs[:, 1, 2], 0.08, atol=5e-4)  # ball rests on the plane
    o.step(act)
    assert (o.reset_buf == 1).all() and (o.rew == 0).all() and (o.timeout_buf == 1).all()
    o.step(act)
    assert (o.progress_buf == 0).all() and (o.reset_buf == 0).all()


def test_ball_drop_roll_and_kick_contact():
    """Ball: free fall matches g t^2/2 before impact, no bounce (restitution 0), rolling without slipping after a
    push, spin decays with the angular damping; a leg box hitting the ball accelerates it (equal and opposite)."""
    o = Oracle(num_envs=1)
    root = o.root_states.reshape(1, 2, 13).copy()
    root[0, 1, 0:3] = [1.0, 0.5, 0.5]; root[0, 1, 7:13] = 0
    o.set_root_states(root.reshape(-1, 13))
    z = []
    for t in range(12):
        o.simulate(); z.append(o.root_states.reshape(1, 2, 13)[0, 1, 2])
    tt = 0.01667 * np.arange(1, 13)
    np.testing.assert_allclose(z, 0.5 - 0.5 * 9.81 * tt * (tt + 0.01667 / 2), atol=2e-4)  # semi-implicit Euler, 2 substeps
    for t in range(60):
        o.simulate()
    r = o.root_states.reshape(1, 2, 13)[0, 1]
    assert abs(r[2] - 0.08) < 5e-4 and abs(r[9]) < 1e-3
    root = o.root_states.reshape(1, 2, 13).copy()
    root[0, 1, 7] = 1.0  # push along x
    o.set_root_states(root.reshape(-1, 13))
    for t in range(60):
        o.simulate()
    r = o.root_states.reshape(1, 2, 13)[0, 1]
    assert 0.3 < r[7] < 1.0 and abs(r[7] - r[11] * 0.08) < 0.02  # v = w R: rolling
    # kick: put the ball just in front of the left foot and swing the hip forward
    o2 = Oracle(num_envs=1)
    for t in range(60):
        o2.step(np.zeros((1, 18), np.float32))
    rb = o2.rigid_body_states.reshape(1, 22, 13)
    foot = rb[0, 12, 0:3]
    root = o2.root_states.reshape(1, 2, 13).copy()
    root[0, 1, 0:3] = [foot[0] + 0.045 + 0.08 + 0.005, foot[1], 0.08]
    o2.set_root_states(root.reshape(-1, 13))
    act = np.zeros((1, 18), np.float32); act[0, 6] = 1.0; act[0, 7] = 0.6  # left thigh forward, knee extend
    vmax, hit = 0.0, False
    for t in range(25):
        o2.step(act)
        cf = o2.contact_forces.reshape(1, 22, 3)[0]
        leg = cf[[8, 9, 10, 11, 12]].sum(axis=0) - np.array([0, 0, cf[12, 2]])
        if np.linalg.norm(cf[21, :2]) > 0.2:
            hit = True
        vmax = max(vmax, o2.root_states.reshape(1, 2, 13)[0, 1, 7])
    assert hit and vmax > 0.2, (hit, vmax)


def test_pd_step_response_and_limits():
    """Arm joint follows a target step without overshoot beyond the limit, respects the 2*pi rad/s clamp, and a
    target outside the limits is clamped to them (kick_env.py:417-418)."""
    o = Oracle(num_envs=1)
    act = np.zeros((1, 18), np.float32)
    act[0, 3] = -3.9  # left forearm: default 1.5, lower limit 0 -> target clamps to 0
    qs, vs = [], []
    for t in range(60):
        o.step(act)
        d = o.dof_state.reshape(1, 18, 2)[0, 3]
        qs.append(d[0]); vs.append(d[1])
        if o.reset_buf[0]:
            break
    assert np.abs(vs).max() <= 2 * np.pi + 1e-6
    assert o.targets[0, 3] == 0.0
    assert min(qs) > -0.05 and qs[-1] < 0.2


# ------------------------------------------------------------------ rigid contact (BEZ_FLAG_HARD_CONTACT)
def _hard_cfg(n, **over):
    c = abi.default_config(n)
    c.flags |= abi.FLAG_HARD_CONTACT
    for k, v in over.items():
        setattr(c, k, v)
    return c


def test_hard_contact_standing_weight_and_no_creep():
    """Rigid contact: the ready pose stands a whole episode, the feet carry exactly the weight, and -- what the compliant
    model cannot do -- under a tilted gravity (tan = 0.06 < mu = 1) the stance feet STICK: they do not creep."""
    o = Oracle(_hard_cfg(2))
    act = np.zeros((2, 18), np.float32)
    for t in range(899):
        o.step(act)
        assert (o.reset_buf == 0).all(), t
    rs = o.root_states.reshape(2, 2, 13)
    assert np.all(np.abs(rs[:, 0, 2] - 0.325) < 0.01) and np.all(np.linalg.norm(rs[:, 0, :2], axis=1) < 0.05)
    assert np.all(np.abs(rs[:, 0, 7:13]) < 2e-3)
    cf = o.contact_forces.reshape(2, 22, 3)
    np.testing.assert_allclose(cf[:, 12, 2] + cf[:, 20, 2], 2.827994 * 9.81, rtol=5e-3)
    np.testing.assert_allclose(rs[:, 1, 2], 0.08, atol=1e-3)   # ball rests ON the plane (ERP leaves no visible penetration)
    # stiction: gravity tilted by atan(0.06) about y (the ball rolls away and ends the episode after ~175 steps: stop before)
    drift = {}
    for hard in (True, False):
        c = _hard_cfg(1) if hard else abi.default_config(1)
        c.gravity[:] = [0.6, 0.0, -9.81]
        o = Oracle(c)
        for t in range(50):
            o.step(np.zeros((1, 18), np.float32))
        x0 = o.rigid_body_states.reshape(1, 22, 13)[0, [12, 20], 0].copy()   # the two feet
        for t in range(100):
            o.step(np.zeros((1, 18), np.float32))
            assert o.reset_buf[0] == 0
        drift[hard] = float(np.abs(o.rigid_body_states.reshape(1, 22, 13)[0, [12, 20], 0] - x0).max())
    assert drift[True] < 1.5e-4, drift             # rigid: the feet stay where they are (< 1.5 um per step of solver residual)
    assert drift[False] > 5 * drift[True], drift   # the regularised-friction model creeps at ~1 cm/s


def test_hard_contact_ball_is_inelastic_and_rolls():
    """Restitution 0 (bez_kick.yaml:16): a dropped ball is at rest one substep after touching down -- no rebound at all;
    pushed, it ends up rolling without slipping at v0 / (1 + I / (m R^2)) = 0.6 v0 (before the angular damping acts)."""
    c = _hard_cfg(1)
    c.ball_ang_damping = 0.0
    o = Oracle(c)
    root = o.root_states.reshape(1, 2, 13).copy()
    root[0, 1, 0:3] = [1.0, 0.5, 0.3]; root[0, 1, 7:13] = 0
    o.set_root_states(root.reshape(-1, 13))
    zs, vz = [], []
    for t in range(40):
        o.simulate(); r = o.root_states.reshape(1, 2, 13)[0, 1]; zs.append(r[2]); vz.append(r[9])
    assert max(vz) < 0.02, max(vz)                      # no rebound (it arrives at 2 m/s; < 1 % of that is the ERP push-out of a sub-mm overlap)
    assert abs(zs[-1] - 0.08) < 1e-3 and abs(vz[-1]) < 1e-4
    root = o.root_states.reshape(1, 2, 13).copy()
    root[0, 1, 7] = 1.0
    o.set_root_states(root.reshape(-1, 13))
    for t in range(30):
        o.simulate()
    r = o.root_states.reshape(1, 2, 13)[0, 1]
    np.testing.assert_allclose(r[7], 0.6, atol=5e-3)     # sliding -> rolling: v = v0 / (1 + 0.00128 / (0.3 * 0.0064))
    np.testing.assert_allclose(r[11] * 0.08, r[7], atol=1e-4)   # v = w R exactly: sticking contact


def _robot_momentum(model, o):
    """sum over links of m (v_origin + w x R c): the oracle's rigid-body rows give every link's origin velocity and spin"""
    from tests.rbd_numpy import quat_to_mat
    rb = o.rigid_body_states.reshape(o.nbe, 13).astype(np.float64)
    p = np.zeros(3)
    for L in model["links"]:
        r = rb[L["body"]]
        p += L["mass"] * (r[7:10] + np.cross(r[10:13], quat_to_mat(r[3:7]) @ np.array(L["com"])))
    return p


def test_hard_contact_ball_robot_impact_is_inelastic(model):
    """Zero gravity, the robot floating at rest, the ball thrown at the torso box at 1 m/s: with rigid contact the two leave
    together (restitution 0: relative normal speed ~ 0) and the robot takes up exactly the momentum the ball lost; the
    compliant model's ball comes back."""
    out = {}
    for hard in (True, False):
        c = _hard_cfg(1) if hard else abi.default_config(1)
        c.gravity[:] = [0.0, 0.0, 0.0]
        o = Oracle(c)
        root = o.root_states.reshape(1, 2, 13).copy()
        root[0, 0, 2] = 5.0
        root[0, 1, 0:3] = [0.30, 0.0, 5.0 - 0.052]; root[0, 1, 7:10] = [-1.0, 0.0, 0.0]
        o.set_root_states(root.reshape(-1, 13))
        p0 = _robot_momentum(model, o)
        for t in range(20):
            o.set_reset(np.zeros(1, np.int64)); o.simulate()
        rs = o.root_states.reshape(1, 2, 13)[0]
        out[hard] = (float(rs[1, 7]), float(rs[0, 7]), _robot_momentum(model, o) - p0, 0.3 * (rs[1, 7:10].astype(np.float64) - [-1.0, 0, 0]))
    vb, vt, dp_robot, dp_ball = out[True]
    assert -0.2 < vb < -0.05 and abs(vb - vt) < 0.03, out            # ball and torso front move on together
    np.testing.assert_allclose(dp_robot, -dp_ball, atol=2e-3)         # what the ball lost (0.27 N s) the robot gained
    assert out[False][0] > 0.0, out                                   # compliant contact: the ball comes back
    np.testing.assert_allclose(out[False][2], -out[False][3], atol=2e-3)


# ------------------------------------------------------------------ TGS-shaped unified substep (BEZ_FLAG_TGS_SOLVER, oracle/bez_oracle_tgs.inc)
def _tgs_cfg(n, tune=None, **over):
    c = abi.default_config(n)
    c.flags |= abi.FLAG_TGS_SOLVER
    for k, v in (tune or {}).items():
        c.tune[k] = v
    for k, v in over.items():
        setattr(c, k, v)
    return c


@pytest.mark.parametrize("tune", [{}, {18: 1.0}, {11: 4.0, 18: 1.0}, {12: 1.0, 8: 8.0}])
def test_tgs_solver_stands_and_carries_the_weight(tune):
    """The unified solver (drives, joint friction, limits, speed limit and contacts as clamped rows over posIters sub-steps): the ready
    pose stands 300 control steps without a reset, the feet carry exactly the robot's weight, the ball rests on the plane, nothing
    blows up -- for the defaults, the compliant leg<->leg contact, the load-proportional joint friction and 8 iterations + speed rows."""
    o = Oracle(_tgs_cfg(2, tune))
    act = np.zeros((2, 18), np.float32)
    cf = np.zeros((2, 22, 3))
    for t in range(300):
        o.step(act)
        assert (o.reset_buf == 0).all(), t
        if t >= 240:
            cf += o.contact_forces.reshape(2, 22, 3) / 60.0   # mean over the last second (joint stick-slip leaves a small vertical ripple)
    rs = o.root_states.reshape(2, 2, 13)
    assert np.isfinite(rs).all() and np.all(np.abs(rs[:, 0, 2] - 0.325) < 0.01)
    # ALL rows (robot + ball): leg <-> leg and ball <-> leg contacts are rigid rows with equal and opposite forces on two bodies, so only
    # the sum over everything is the ground reaction = the weight of robot and ball (the default variant creeps forward by ~6 mm/s --
    # its friction anchors do not persist across substeps -- and ends up leaning on the ball)
    np.testing.assert_allclose(cf[:, :, 2].sum(1), (2.827994 + 0.3) * 9.81, rtol=2e-2)
    np.testing.assert_allclose(rs[:, 1, 2], 0.08, atol=1e-3)
    ds = o.dof_state.reshape(2, 18, 2)
    assert np.abs(ds[:, :, 0] - o.targets).max() < 0.05   # the drives hold the pose (knee / thigh sag ~0.02 rad under the weight)


def test_tgs_drive_is_the_first_order_response_of_the_yaml_gains():
    """An unloaded joint under the TGS drive row follows the same law as the implicit PD of the compliant model: rate = Kp / Kd x error
    (13.3 1/s at the yaml's 100 / 7.5) -- the head joint, target stepped by 0.2 rad, covers 1 - exp(-dt Kp / Kd) of the way per step."""
    c = _tgs_cfg(1, {18: 1.0})
    o = Oracle(c)
    act = np.zeros((1, 18), np.float32)
    for _ in range(60):
        o.step(act)
    q0 = o.dof_state.reshape(18, 2)[2, 0]          # left arm joint 0 (the head's targets are forced to 0 by pre_physics_step)
    act[0, 2] = 0.2
    errs = []
    for _ in range(6):
        o.step(act)
        errs.append(o.targets[0, 2] - o.dof_state.reshape(18, 2)[2, 0])
    assert abs(o.targets[0, 2] - (q0 + 0.2)) < 0.02
    ratio = np.array(errs[1:]) / np.array(errs[:-1])
    np.testing.assert_allclose(ratio, np.exp(-float(c.dt) * 100.0 / 7.5), atol=0.06)


def test_tgs_flag_is_oracle_only():
    """(the HIP library refuses it: tests/test_abi_cpu.py) -- here: the oracle's TGS and compliant models really are different models"""
    a, b = Oracle(_tgs_cfg(1)), Oracle(abi.default_config(1))
    act = np.full((1, 18), 0.3, np.float32)
    for _ in range(20):
        a.step(act); b.step(act)
    assert np.abs(a.root_states - b.root_states).max() > 1e-4


def test_ankle_stop_is_the_calf_foot_plate_contact(model):
    """BEZ_FLAG_ANKLE_STOP (oracle only; kick_env.py:365-366 collision_filter 0, soccerbot_stl.urdf:232-236 / 272-276): the gap between the calf
    box's bottom corners and the foot plate's top face is a function of ankle pitch and foot roll alone.  tools/arm_posture.ankle_gap derives it
    in Python from the baked model (axes, origins, boxes); the oracle's contact must be active exactly where that gap is negative, act on the
    calf and foot rows with equal and opposite forces along the foot's z axis, and push the two ankle joints back out of the contact."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from arm_posture import ankle_gap
    cfg = abi.default_config(1)
    cfg.flags |= abi.FLAG_ANKLE_STOP
    cfg.gravity[:] = [0.0, 0.0, 0.0]
    o = Oracle(cfg)
    plain = Oracle(abi.default_config(1))
    dflt = np.asarray(model["dof_default"], float)
    calf_l, foot_l, calf_r, foot_r = 10, 12, 18, 20           # Isaac body rows (bez_model.json body_names)
    cases = [(0.613, 0.0), (0.79, 0.23), (0.79, 0.45), (1.2, 0.6), (0.3, -0.75), (1.3, -0.5), (0.0, 0.78)]
    seen_active = seen_free = 0
    for side, (ip, ir, calf, foot) in (("left", (8, 9, calf_l, foot_l)), ("right", (16, 17, calf_r, foot_r))):
        for pitch, roll in cases:
            q = dflt.copy(); q[ip], q[ir] = pitch, roll
            for orc in (o, plain):
                _inject(orc, [0.0, 0.0, 1.0], [0, 0, 0, 1], np.zeros(6), q, np.zeros(18))   # in the air, at rest: no ground, no drive error but the pose's
                orc.set_targets(q[None].astype(np.float32))
            gap = ankle_gap(model, side, pitch, roll)
            _, qdd, _, cf = o.forward_dynamics(0, mode=0)
            _, qdd0, _, cf0 = plain.forward_dynamics(0, mode=0)
            f_calf, f_foot = cf[calf], cf[foot]
            if gap < -2e-4:
                seen_active += 1
                assert np.linalg.norm(f_foot) > 1.0, (side, pitch, roll, gap, f_foot)
                np.testing.assert_allclose(f_calf, -f_foot, rtol=1e-9, atol=1e-9)        # an internal force pair
                # along the foot's z axis (foot frame = calf frame rotated by the two ankle joints; the torso is upright and the hip joints are at 0)
                assert abs(np.linalg.norm(f_foot)) == pytest.approx(2e5 * -gap, rel=0.35), (gap, f_foot)   # k * depth of the deepest corner (others may add)
                # the stop opens the contact: it drives the roll joint back towards a smaller |roll|
                assert (qdd[ir] - qdd0[ir]) * np.sign(roll) < 0, (side, pitch, roll, qdd[ir], qdd0[ir])
            elif gap > 2e-4:
                seen_free += 1
                np.testing.assert_array_equal(cf, cf0)
                np.testing.assert_allclose(qdd, qdd0, rtol=0, atol=1e-12)
    assert seen_active >= 6 and seen_free >= 4
