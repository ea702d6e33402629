"""Round 6 on the GPU: the in-dynamics joint speed limit and the leg <-> leg contact with its common implicit
scale, on BOTH step kernels through the C ABI -- the same known answers as tests/test_oracle_round6.py, plus HIP = oracle on the scenario."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from bez_isaacgym_amd import abi  # noqa: E402
from tests import rbd_numpy as R  # noqa: E402
from tests.scenarios import capsule_penetration, leg_press  # noqa: E402
from tests.test_oracle_round6 import _free_space_cfg, _pressed_state  # noqa: E402


def _adapter(cfg, kernel, monkeypatch):
    from tests.sim_adapter import SimAdapter
    monkeypatch.setenv("BEZ_SIM_KERNEL", kernel)
    return SimAdapter(cfg)


def _inject(sim, n, model, seed):
    """n pressed states (tests/test_oracle_round6._pressed_state) through the Isaac-layout setters; returns them"""
    rng = np.random.default_rng(seed)
    dflt = np.asarray(model["dof_default"], float)
    rs = sim.root_states.reshape(n, 2, 13).copy()
    ds = np.zeros((n, 18, 2), np.float32)
    acts = np.zeros((n, 18), np.float32)
    states = []
    for e in range(n):
        q, qd, target, v0, quat, fast, sign = _pressed_state(model, rng)
        rs[e, 0, 0:3] = (0.0, 0.0, 1.0); rs[e, 0, 3:7] = quat; rs[e, 0, 7:10] = v0[3:]; rs[e, 0, 10:13] = v0[:3]
        rs[e, 1, :] = 0; rs[e, 1, 0:3] = (0.0, 3.0, 0.08); rs[e, 1, 6] = 1.0
        ds[e, :, 0] = q; ds[e, :, 1] = qd
        acts[e] = (target - dflt).astype(np.float32)
        states.append((q, qd, v0, quat))
    sim.set_root_states(rs.reshape(-1, 13)); sim.set_dof_state(ds.reshape(-1, 2))
    return acts, states


@pytest.mark.parametrize("kernel", ["ws8", "ws8q", "lane"])
def test_one_substep_with_locked_joints_needs_no_base_wrench(model, kernel, monkeypatch):
    """One substep (substeps = 1) from the pressed state on the GPU: the accelerations the state change implies, put into the
    independent RNEA, need no wrench on the floating base (fp32 state read back: 2e-2 N / N m of joint torques of 2.5 N m, contact
    forces of tens of newtons and a constraint torque per locked joint), >= 3 joints per env end ON the limit, and the contact rows
    show the legs pressed together."""
    n = 64
    cfg = _free_space_cfg(n, substeps=1)
    sim = _adapter(cfg, kernel, monkeypatch)
    sim.step(np.zeros((n, 18), np.float32))
    acts, states = _inject(sim, n, model, 21)
    rs0 = sim.root_states.reshape(n, 2, 13).astype(np.float64); ds0 = sim.dof_state.reshape(n, 18, 2).astype(np.float64)
    sim.pre_physics(acts); sim.simulate()
    rs1 = sim.root_states.reshape(n, 2, 13).astype(np.float64); ds1 = sim.dof_state.reshape(n, 18, 2).astype(np.float64)
    cf = sim.contact_forces.reshape(n, -1, 3)
    h = float(cfg.dt)
    on = 0; worst = 0.0
    for e in range(n):
        q, qd = ds0[e, :, 0], ds0[e, :, 1]
        quat = rs0[e, 0, 3:7]; w0, v0 = rs0[e, 0, 10:13], rs0[e, 0, 7:10]
        qdd = (ds1[e, :, 1] - qd) / h
        wdot = (rs1[e, 0, 10:13] - w0) / h
        vdot = (rs1[e, 0, 7:10] - v0) / h
        a0 = np.concatenate([wdot, vdot - np.cross(w0, v0)])          # classical -> spatial acceleration of the torso origin (substep: oracle)
        f0, _ = R.rnea_floating(model, quat / np.linalg.norm(quat), np.concatenate([w0, v0]), a0, q, qd, qdd, np.zeros(3))
        worst = max(worst, float(np.abs(f0).max()))
        on += int((np.abs(np.abs(ds1[e, :, 1]) - float(cfg.vel_limit)) < 2e-4).sum() >= 3)
    assert worst < 2e-2, worst
    assert on >= n // 2, on
    assert (np.abs(cf[:, :21]).sum((1, 2)) > 1.0).mean() > 0.5


@pytest.mark.parametrize("kernel", ["ws8", "ws8q", "lane"])
def test_leg_press_hip_equals_oracle_and_stays_bounded(model, kernel, monkeypatch):
    """The scenario that broke the round-5 model, on the GPU: finite, momentum drift within the integrator's bound of the oracle test, and
    over the first control steps (before fp32 / fp64 trajectories of a chaotic scenario part) the same joint angles as the oracle."""
    from tests.scenarios import make_backend
    n = 64
    cfg = _free_space_cfg(n)
    sim = _adapter(cfg, kernel, monkeypatch)
    sim.step(np.zeros((n, 18), np.float32))
    r = leg_press(sim, n, model)
    assert r["finite"] and r["on_limit"] > 5000, r
    assert r["dp"] < 1.5 and r["dL"] < 0.25, r
    orc = make_backend("oracle", _free_space_cfg(4)); orc.step(np.zeros((4, 18), np.float32))
    gpu = _adapter(_free_space_cfg(4), kernel, monkeypatch); gpu.step(np.zeros((4, 18), np.float32))
    a = leg_press(orc, 4, model, steps=6); b = leg_press(gpu, 4, model, steps=6)
    np.testing.assert_allclose(gpu.dof_state.reshape(4, 18, 2)[:, :, 0], orc.dof_state.reshape(4, 18, 2)[:, :, 0], atol=2e-2)
    assert abs(a["penetration"] - b["penetration"]) < 2e-3, (a, b)


@pytest.mark.parametrize("kernel", ["ws8", "ws8q", "lane"])
def test_self_contact_holds_the_legs_apart_hip(model, kernel, monkeypatch):
    """tests/test_oracle_round6.test_self_contact_holds_the_legs_apart on the GPU: steady, < 5 mm, the hip stopped by the other foot."""
    n = 64
    cfg = _free_space_cfg(n)
    o = _adapter(cfg, kernel, monkeypatch)
    o.step(np.zeros((n, 18), np.float32))
    dflt = np.asarray(model["dof_default"], np.float32)
    rs = o.root_states.reshape(n, -1, 13).copy(); rs[:, 0, :] = 0; rs[:, 0, 2] = 1.0; rs[:, 0, 6] = 1.0
    rs[:, 1, :] = 0; rs[:, 1, 0:3] = (0.0, 3.0, 0.08); rs[:, 1, 6] = 1.0
    o.set_root_states(rs.reshape(-1, 13))
    ds = np.zeros((n, 18, 2), np.float32); ds[:, :, 0] = dflt
    o.set_dof_state(ds.reshape(-1, 2))
    act = np.zeros((n, 18), np.float32); act[:, 5] = -0.7
    pens, rolls = [], []
    for k in range(120):
        o.pre_physics(act); o.simulate()
        if k >= 60:
            pens.append(capsule_penetration(o, n, model).max()); rolls.append(o.dof_state.reshape(n, 18, 2)[:, 5, 0].copy())
    assert max(pens) < 0.005, max(pens)
    assert np.ptp(np.array(rolls[30:]), axis=0).max() < 0.01
    assert -0.25 < rolls[-1].min() and rolls[-1].max() < -0.03, (rolls[-1].min(), rolls[-1].max())


def test_in_launch_gradient_norm_on_a_capped_grid_and_its_residency_guard():
    """Round-5 advisor finding (medium): the optimiser launch's in-launch norm is a grid barrier.  (a) bez_ppo_adam_grid_capacity reports
    what the device can hold at once and what the launch uses; on a whole MI355X the 256-workgroup cap fits.  (b) n > 256 * 2048 parameters:
    the grid is capped at 256 workgroups and every workgroup walks several slices -- the step must equal the norm-shares path there too (the
    round-5 test stopped at 61 workgroups).  (c) the counter is back at zero: the buffer serves the next launch."""
    import ctypes as C
    import torch
    from bez_isaacgym_amd.ppo import fused as F
    DEV = "cuda:0"
    cap, g = C.c_int32(0), C.c_int32(0)
    assert F.lib().bez_ppo_adam_grid_capacity(124237, C.addressof(cap), C.addressof(g)) == 0
    assert g.value == 61 and cap.value >= 256 and F.adam_grid_fits(124237)
    n = 256 * 2048 * 3 + 1234 + 1
    assert F.lib().bez_ppo_adam_grid_capacity(n, C.addressof(cap), C.addressof(g)) == 0 and g.value == 256 and F.adam_grid_fits(n)
    torch.manual_seed(3)
    gs = (torch.randn(n + 3, device=DEV)[:n] * 3.0 * 1024.0).contiguous()
    p0 = torch.randn(n, device=DEV)
    grid_buf = torch.zeros(F.ADAM_GRIDNORM_FLOATS, device=DEV)

    def step(grid):
        p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        steps, lr = torch.zeros(1, device=DEV), torch.tensor([3e-4], device=DEV)
        scale, gt = torch.tensor([1024.0], device=DEV), torch.zeros(1, device=DEV, dtype=torch.int32)
        work = torch.zeros(F.ADAM_WORK_FLOATS, device=DEV)
        parts = None if grid else F.grad_norm_parts(gs, torch.zeros((n // 4 + 3 + 1023) // 1024 + 1, 2, device=DEV))
        F.adam_step(p, gs, m, v, steps, lr, (0.9, 0.999), 1e-8, 0.0, 1.0, scale, gt, 2.0, 0.5, 2000, work, norm_parts=parts,
                    grid_norm=grid_buf if grid else None)
        torch.cuda.synchronize()
        return p, m
    p_ref, m_ref = step(False)
    for _ in range(2):
        p_g, m_g = step(True)
        # (the two norms add 1.5 M squares in different fixed orders: the clip coefficient may differ in its last bit, an ulp in 0.1 % of the elements)
        torch.testing.assert_close(p_g, p_ref, rtol=3e-7, atol=2e-9); torch.testing.assert_close(m_g, m_ref, rtol=1e-6, atol=1e-9)
        assert int(grid_buf[512:513].view(torch.int32)) == 0
    assert not torch.equal(p_ref, p0)


def test_scenario_contact_variants_step_like_the_oracle(model):
    """BEZ_FLAG_ALL_GROUND_SHAPES and BEZ_FLAG_ANKLE_STOP on the GPU (one-env-per-lane kernel), no longer rc -5.
    Resynchronised single-step parity against the fp64 oracle on states that exercise both: half the envs lying on their back / side
    with bent legs (hip, thigh, calf and forearm corners on the ground), ankles flexed and rolled into the calf <-> foot-plate stop."""
    from oracle.bez_oracle import Oracle
    from tests.parity_util import EnvOutliers
    from tests.sim_adapter import SimAdapter
    n = 128
    fl = abi.FLAG_IMU_PREV_ALIAS | abi.FLAG_ALL_GROUND_SHAPES | abi.FLAG_ANKLE_STOP
    co, cg = abi.default_config(n, seed=9), abi.default_config(n, seed=9)
    co.flags = fl; cg.flags = fl
    o, g = Oracle(co), SimAdapter(cg)
    plain = Oracle(abi.default_config(n, seed=9))
    rng = np.random.default_rng(4)
    dflt = np.asarray(model["dof_default"], np.float32)
    rs = o.root_states.reshape(n, 2, 13).copy()
    S = np.sqrt(0.5)
    for e in range(n // 2):                      # lying: back, front or side, just above the ground
        rs[e, 0, 2] = 0.09
        rs[e, 0, 3:7] = [(0.0, -S, 0.0, S), (0.0, S, 0.0, S), (S, 0.0, 0.0, S)][e % 3]
    rs[:, 1, 0:3] = (0.0, 2.0, 0.08)
    ds = o.dof_state.reshape(n, 18, 2).copy()
    ds[:, :, 0] = dflt + rng.uniform(-0.3, 0.3, (n, 18)).astype(np.float32)
    ds[n // 2:, 8, 0] = rng.uniform(0.7, 1.3, n - n // 2); ds[n // 2:, 9, 0] = rng.uniform(0.3, 0.7, n - n // 2) * rng.choice([-1, 1], n - n // 2)      # left ankle into the stop
    ds[n // 2:, 16, 0] = rng.uniform(0.7, 1.3, n - n // 2); ds[n // 2:, 17, 0] = rng.uniform(0.3, 0.7, n - n // 2) * rng.choice([-1, 1], n - n // 2)
    for x in (o, plain):
        x.set_root_states(rs.reshape(-1, 13)); x.set_dof_state(ds.reshape(-1, 2))
    T = EnvOutliers(n, share=4e-3)
    differs = 0.0
    for t in range(20):
        g.set_root_states(o.root_states); g.set_dof_state(o.dof_state); g.set_contact_forces(o.contact_forces); g.set_targets(o.targets)
        g.set_reset(o.reset_buf); g.set_progress(o.progress_buf)
        plain.set_root_states(o.root_states); plain.set_dof_state(o.dof_state); plain.set_targets(o.targets)
        act = rng.uniform(-0.5, 0.5, (n, 18)).astype(np.float32)
        for x in (o, g, plain):
            x.pre_physics(act); x.simulate()
        do, dg = o.dof_state.reshape(n, 18, 2), g.dof_state.reshape(n, 18, 2)
        T.close(dg[..., 0], do[..., 0], 2e-4, what="q"); T.close(dg[..., 1], do[..., 1], 3e-2, what="qd")
        ro, rg = o.root_states.reshape(n, 2, 13), g.root_states.reshape(n, 2, 13)
        T.close(rg[:, 0, 0:7], ro[:, 0, 0:7], 5e-5, what="pose"); T.close(rg[:, 0, 7:13], ro[:, 0, 7:13], 1e-2, what="vel")
        if t > 0:   # (step 0: the ball has just been teleported onto the ground 2 m away; its row holds that transient)
            T.close(g.contact_forces, o.contact_forces, 0.08, rtol=0.02, what="cf")
        T.end_step()
        differs = max(differs, float(np.abs(do[..., 1] - plain.dof_state.reshape(n, 18, 2)[..., 1]).max()))
    T.finish()
    assert differs > 0.5, differs                          # the two variants really change the step (against the plain model on the same states)
    cf = g.contact_forces.reshape(n, 22, 3)
    assert np.abs(cf[: n // 2, [6, 7, 8, 9, 10, 14, 15, 16, 17, 18]]).sum() > 10.0   # forearm / hip / thigh / calf rows loaded in the lying envs


@pytest.mark.gpu
@pytest.mark.parametrize("substeps", [1, 3, 4])
def test_lane_group_kernel_at_other_substep_counts(substeps, monkeypatch):
    """The lane-group kernel's hand-overs are ordered by sequence words that count substeps (leg <-> leg sums, link packages): the default
    yaml runs 2; 1, 3 and 4 substeps step like the one-lane kernel too (resynchronised every step, the usual bars and outlier bookkeeping)."""
    from tests.parity_util import EnvOutliers
    from tests.test_tasks import make_cfg
    n = 120

    def cfg():
        c = make_cfg(n, seed=17)
        c.substeps = substeps
        return c
    a = _adapter(cfg(), "ws8", monkeypatch)
    b = _adapter(cfg(), "ws8q", monkeypatch)
    rng = np.random.default_rng(5)
    T = EnvOutliers(n)
    for t in range(25):
        b.set_root_states(a.root_states); b.set_dof_state(a.dof_state); b.set_contact_forces(a.contact_forces)
        b.set_targets(a.targets); b.set_reset(a.reset_buf); b.set_progress(a.progress_buf); b.set_prev_lin_vel(a.prev_lin_vel)
        act = rng.uniform(-1, 1, (n, 18)).astype(np.float32)
        a.step(act); b.step(act)
        np.testing.assert_array_equal(b.progress_buf, a.progress_buf)
        ra, rb = a.root_states, b.root_states
        assert np.isfinite(rb).all()
        T.close(rb[..., 0:7], ra[..., 0:7], 2e-5, what="pose")
        T.close(rb[..., 7:13], ra[..., 7:13], 4e-3, what="vel")
        da, db = a.dof_state.reshape(n, 18, 2), b.dof_state.reshape(n, 18, 2)
        T.close(db[..., 0], da[..., 0], 1e-4, what="q")
        T.close(db[..., 1], da[..., 1], 1.5e-2, what="qd")
        T.close(b.contact_forces, a.contact_forces, 2.5e-2, rtol=4e-3, what="cf")
        T.end_step()
    T.finish()
