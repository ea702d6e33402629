"""Round-3 GPU tests (through the C ABI): HIP-graph replay safety of the bez_walk / bez_orient goal draw, ...
All tests need a GPU: `pytest -m gpu`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("task", ["bez_walk", "bez_orient"])
def test_walk_goal_draw_advances_under_graph_replay(task):
    """the goal of bez_walk / bez_orient is drawn per reset call (walk_env.py:570-575).  A captured HIP
    graph replays kernel arguments verbatim, so the draw must come from device state: capture ONE fused step, replay it with all
    envs flagged for reset, and the goal must follow the oracle's call-counter sequence step by step."""
    import torch
    from bez_isaacgym_amd import abi
    from tests.sim_adapter import SimAdapter
    from tests.test_tasks import make_cfg, oracle
    n = 96
    g = SimAdapter(make_cfg(n, task=task, seed=5))
    o = oracle(n, task=task, seed=5)
    act = torch.zeros(n * 18, device=g.dev)
    reset = g.sim.tensor(abi.TENSOR_RESET)
    zero = np.zeros((n, 18), np.float32)
    # two eager steps (as the PPO loop's warm-up), then capture
    for _ in range(2):
        reset.fill_(1); o.set_reset(np.ones(n, np.int64))
        g.sim.step(act); o.step(zero)
        np.testing.assert_array_equal(g.goal, o.goal)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        reset.fill_(1)
        g.sim.step(act)  # warm the stream
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            g.sim.step(act)
    o.set_reset(np.ones(n, np.int64)); o.step(zero)   # the warm-up step above
    # NOTE: capture itself does not execute the step
    seen = []
    for k in range(6):
        reset.fill_(1); o.set_reset(np.ones(n, np.int64))
        graph.replay(); torch.cuda.synchronize()
        o.step(zero)
        gg = g.goal
        np.testing.assert_array_equal(gg, o.goal)      # bit-exact: same Philox draw, same counter
        assert (gg == gg[0]).all()                       # one draw per call, shared by every env it resets
        seen.append(tuple(gg[0]))
    assert len(set(seen)) == 6, seen                     # a fresh goal on every replay


def _dr_cfg(freq=5, sched=40):
    from bez_isaacgym_amd import abi
    r = lambda a, b, s=sched: {"range": [a, b], "schedule": "linear", "schedule_steps": s}
    return abi.dr_config_from_params({
        "frequency": freq,
        "observations": {"range": [0, .002], "operation": "additive", "distribution": "gaussian"},
        "actions": {"range": [0., .02], "operation": "additive", "distribution": "gaussian", "schedule": "linear", "schedule_steps": sched},
        "sim_params": {"gravity": dict(r(0, 0.4), operation="additive", distribution="gaussian")},
        "actor_params": {"bez": {
            "rigid_shape_properties": {"friction": dict(r(0.7, 1.3), num_buckets=500, operation="scaling", distribution="uniform")},
            "dof_properties": {"damping": dict(r(0.5, 1.5), operation="scaling", distribution="uniform"),
                               "stiffness": dict(r(0.5, 1.5), operation="scaling", distribution="uniform"),
                               "lower": dict(r(0, 0.01), operation="additive", distribution="gaussian"),
                               "upper": dict(r(0, 0.01), operation="additive", distribution="gaussian")}}}})


def test_device_side_domain_randomization_matches_oracle():
    """the redraw at reset time (vec_task.py:505-725 via kick_env.py:781-782) runs inside the simulator.
    Same keyed Philox draws in the oracle: randomize_buf, the reset flags and every uniform-derived array (friction buckets,
    Kp, Kd) bit-exact; Box-Muller arrays (limits, gravity) and the noise scalars to 2e-6 (libm vs device logf / cosf); and the
    physics of the randomised envs keeps the usual parity bars.  N is not a multiple of 1024 (the DR kernel's stride)."""
    from bez_isaacgym_amd import abi
    from oracle.bez_oracle import Oracle
    from tests.sim_adapter import SimAdapter
    n = 1500
    dr = _dr_cfg()
    o, g = Oracle(abi.default_config(n, seed=9, env_id_offset=4096)), SimAdapter(abi.default_config(n, seed=9, env_id_offset=4096))
    o.set_randomization(dr); g.set_randomization(dr)
    exact = (abi.PARAM_FRICTION, abi.PARAM_KP_SCALE, abi.PARAM_KD_SCALE)
    close = (abi.PARAM_DOF_LOWER, abi.PARAM_DOF_UPPER, abi.PARAM_GRAVITY)

    def compare():
        np.testing.assert_array_equal(g.randomize_buf, o.randomize_buf)
        for p in exact:
            np.testing.assert_array_equal(g.get_env_params(p), o.get_env_params(p))
        for p in close:
            np.testing.assert_allclose(g.get_env_params(p), o.get_env_params(p), rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(g.dr_noise, o.dr_noise, rtol=1e-6, atol=0)
    compare()
    np.testing.assert_array_equal(o.get_env_params(abi.PARAM_FRICTION), 1.0)   # frame 0 of a linear schedule: nothing randomised yet
    rng = np.random.default_rng(2)
    redraws, grav = 0, set()
    for t in range(60):
        g.set_root_states(o.root_states); g.set_dof_state(o.dof_state); g.set_contact_forces(o.contact_forces)
        g.set_targets(o.targets); g.set_reset(o.reset_buf); g.set_progress(o.progress_buf)
        for p in close:   # keep the 1-ulp differences of the Box-Muller arrays out of the physics comparison
            g.set_env_params(p, o.get_env_params(p))
        before = o.get_env_params(abi.PARAM_KP_SCALE).copy()
        act = rng.uniform(-1, 1, (n, 18)).astype(np.float32)
        o.step(act); g.step(act)
        compare()
        redraws += int((np.abs(o.get_env_params(abi.PARAM_KP_SCALE) - before).max(axis=1) > 0).sum())
        grav.add(tuple(np.round(o.get_env_params(abi.PARAM_GRAVITY)[0], 6)))
        do, dg = o.dof_state.reshape(n, 18, 2), g.dof_state.reshape(n, 18, 2)
        flipped = np.abs(dg[..., 0] - do[..., 0]).max(1) > 1.5e-3   # an env on the other side of a switch (counted below) may also fall a step earlier / later
        np.testing.assert_array_equal(g.reset_buf[~flipped], o.reset_buf[~flipped])
        # the physics under the redrawn parameters: the usual bars for all but a handful of knife-edge elements (a jittered joint
        # limit or a saturating drive within rounding of its switch: DESIGN.md 6 "knife edges"), which stay small
        eq, ev = np.abs(dg[..., 0] - do[..., 0]), np.abs(dg[..., 1] - do[..., 1])
        assert (eq < 1.5e-4).mean() > 0.999 and eq.max() < 0.5, (t, eq.max())   # (round 6: a speed-limit switch within rounding is a larger jump than a saturating drive's)
        assert (ev < 1.5e-2).mean() > 0.999, (t, ev.max())
    assert redraws > n // 4, redraws          # random actions: most envs fell and were redrawn at least once
    assert len(grav) >= 5, grav               # gravity refreshed every `frequency` frames in which some env reset
    f = o.get_env_params(abi.PARAM_FRICTION)
    assert f.min() < 0.9 and f.max() > 1.1    # schedule complete: the whole U(0.7, 1.3) range is reached
    assert o.dr_noise[3] == np.float32(0.02) and o.dr_noise[1] == np.float32(0.002)


def test_device_side_domain_randomization_is_shard_invariant():
    """the draws are keyed by the GLOBAL env id: two shards of 300 envs reproduce the arrays of one sim of 600"""
    from bez_isaacgym_amd import abi
    from tests.sim_adapter import SimAdapter
    dr = _dr_cfg(freq=1, sched=0)
    whole = SimAdapter(abi.default_config(600, seed=4))
    parts = [SimAdapter(abi.default_config(300, seed=4, env_id_offset=300 * k)) for k in range(2)]
    for x in [whole] + parts:
        x.set_randomization(dr)
    for p in (abi.PARAM_FRICTION, abi.PARAM_KP_SCALE, abi.PARAM_DOF_UPPER):
        np.testing.assert_array_equal(np.concatenate([x.get_env_params(p) for x in parts]), whole.get_env_params(p))


def test_dr_noise_kernel_statistics_and_schedule():
    """bez_sim_add_dr_noise = the gaussian additive noise lambda of vec_task.py:586-592 in one launch: mean / std follow the
    device-resident BEZ_TENSOR_DR_NOISE, draws differ between calls of different frames and between observation / action kind."""
    import torch
    from bez_isaacgym_amd import abi
    from tests.sim_adapter import SimAdapter
    n = 4096
    g = SimAdapter(abi.default_config(n, seed=1))
    g.set_randomization(_dr_cfg(freq=1, sched=0))
    x = torch.zeros(n * 54, device=g.dev)
    g.sim.add_dr_noise(x, 0)
    assert abs(float(x.mean())) < 2e-5 and abs(float(x.std()) - 0.002) < 2e-5           # observations: N(0, 0.002)
    a = torch.zeros(n * 18 + 3, device=g.dev)                                            # not a multiple of 4
    g.sim.add_dr_noise(a, 1)
    assert abs(float(a.std()) - 0.02) < 3e-4 and float(a[-3:].abs().min()) > 0
    k = torch.mean(((a - a.mean()) / a.std()) ** 4)
    assert abs(float(k) - 3.0) < 0.15                                                    # gaussian kurtosis
    b = torch.zeros_like(a)
    g.sim.add_dr_noise(b, 1)
    assert torch.equal(a, b)                 # same frame, same kind: the same stream (one call per step and kind)
    g.step(np.zeros((n, 18), np.float32))    # next frame
    c = torch.zeros_like(a)
    g.sim.add_dr_noise(c, 1)
    assert not torch.equal(a, c) and abs(float((a * c).mean())) < 2e-5


@pytest.mark.parametrize("kernel,n", [("ws8", 4096), ("ws8", 200), ("ws8q", 200), ("lane", 256)])
def test_observation_noise_inside_the_step_equals_the_separate_launch(kernel, n, monkeypatch):
    """BEZ_FLAG_OBS_NOISE_IN_STEP: the post-physics part writes the noisy observations itself (the noise rides on the kernel's
    copy-out).  Two simulators with the same seed and randomisation, one with the flag, one adding the noise with
    bez_sim_add_dr_noise after every step: bit-identical observations over 40 steps (resets, a partial last workgroup at n = 200),
    everything else identical too; the call on the observation tensor is a no-op with the flag, a real launch on any other tensor."""
    import torch
    from bez_isaacgym_amd import abi
    from tests.sim_adapter import SimAdapter
    monkeypatch.setenv("BEZ_SIM_KERNEL", kernel)
    cfg_a, cfg_b = abi.default_config(n, seed=9), abi.default_config(n, seed=9)
    cfg_a.flags |= abi.FLAG_OBS_NOISE_IN_STEP
    a, b = SimAdapter(cfg_a), SimAdapter(cfg_b)
    for g in (a, b):
        g.set_randomization(_dr_cfg(freq=5, sched=0))
    obs_a, obs_b = a.sim.tensor(abi.TENSOR_OBS), b.sim.tensor(abi.TENSOR_OBS)
    rng = np.random.default_rng(2)
    clean_differs = False
    for t in range(40):
        act = rng.uniform(-1, 1, (n, 18)).astype(np.float32)
        a.step(act); b.step(act)
        clean = obs_b.clone()
        a.sim.add_dr_noise(obs_a, 0)          # no-op: the kernel has added it
        b.sim.add_dr_noise(obs_b, 0)          # the separate launch
        assert torch.equal(obs_a, obs_b), (t, float((obs_a - obs_b).abs().max()))
        clean_differs |= not torch.equal(clean, obs_b)
        d = (obs_b - clean).flatten()
        assert abs(float(d.std()) - 0.002) < 2e-4 and abs(float(d.mean())) < 2e-4
    assert clean_differs
    np.testing.assert_array_equal(a.root_states, b.root_states)
    np.testing.assert_array_equal(a.rew, b.rew); np.testing.assert_array_equal(a.reset_buf, b.reset_buf)
    x = torch.zeros(n * 54, device=a.dev)
    a.sim.add_dr_noise(x, 0)
    assert float(x.std()) > 1e-3              # any other tensor still gets its noise


def test_lean_step_changes_nothing_the_rollout_reads():
    """BEZ_FLAG_LEAN_STEP: the fused step skips the stores of the contact-force rows, FEET and PREV_LIN_VEL (308 B / env-step).
    State, observations (incl. the feet flags, computed from the in-kernel forces), reward and bookkeeping stay bit-identical;
    the three tensors freeze at their last non-lean values."""
    from bez_isaacgym_amd import abi
    from tests.sim_adapter import SimAdapter
    n = 300
    a = SimAdapter(abi.default_config(n, seed=5))
    c = abi.default_config(n, seed=5); c.flags |= abi.FLAG_LEAN_STEP
    b = SimAdapter(c)
    rng = np.random.default_rng(1)
    frozen = None
    for t in range(40):
        act = rng.uniform(-1, 1, (n, 18)).astype(np.float32)
        a.step(act); b.step(act)
        np.testing.assert_array_equal(b.obs, a.obs)
        np.testing.assert_array_equal(b.rew, a.rew)
        np.testing.assert_array_equal(b.reset_buf, a.reset_buf); np.testing.assert_array_equal(b.progress_buf, a.progress_buf)
        np.testing.assert_array_equal(b.root_states, a.root_states); np.testing.assert_array_equal(b.dof_state, a.dof_state)
        np.testing.assert_array_equal(b.targets, a.targets)
        if t == 0:
            np.testing.assert_array_equal(b.contact_forces, a.contact_forces)   # the process's first step is never lean (prev_lin_vel is read back once)
            frozen = b.contact_forces
        else:
            np.testing.assert_array_equal(b.contact_forces, frozen)
    assert np.abs(a.contact_forces - frozen).max() > 1.0
    assert a.obs[:, 44:52].max() == 1.0 and a.obs[:, 44:52].min() == -1.0


@pytest.mark.parametrize("kernel", ["ws8", "ws8q", "lane"])
def test_free_flight_momentum_at_full_size(model, kernel, monkeypatch):
    """A size-independent property at BASELINE's full size (4096 envs) on the production kernel: in free flight (robot dropped from
    5 m, the ball parked far away) drives, joint friction and joint limits are INTERNAL forces -- whatever the 18 targets do, the
    robot's linear momentum changes by exactly M g t and its angular momentum about the centre of mass does not change.  The sums
    come from the Isaac-visible rigid-body rows (origin velocity, spin) and the URDF masses / inertias, i.e. independently of the
    kernel's own articulated-body quantities.  (Round 6: the speed limit itself is a constraint inside the ABA and reacts on the parent --
    tests/test_gpu_round6.py holds that; here small actions keep the motion slow enough for the integrator's own drift to stay under the bar.)"""
    import torch
    from bez_isaacgym_amd import abi
    from tests.rbd_numpy import quat_to_mat
    from tests.sim_adapter import SimAdapter
    monkeypatch.setenv("BEZ_SIM_KERNEL", kernel)
    n = 4096
    cfg = abi.default_config(n, seed=11)
    g = SimAdapter(cfg)
    root = g.root_states.reshape(n, 2, 13).copy()
    root[:, 0, 2] = 5.0
    root[:, 1, 0:3] = [50.0, 50.0, 0.08]
    g.set_root_states(root.reshape(-1, 13))
    links = model["links"]
    mass = np.array([L["mass"] for L in links]); com = np.array([L["com"] for L in links]); body = np.array([L["body"] for L in links])
    inertia = np.array([[[L["inertia"][0], L["inertia"][3], L["inertia"][4]], [L["inertia"][3], L["inertia"][1], L["inertia"][5]],
                         [L["inertia"][4], L["inertia"][5], L["inertia"][2]]] for L in links])
    M = mass.sum()

    def momenta():
        rb = g.rigid_body_states.reshape(n, -1, 13).astype(np.float64)[:, body]            # (n, links, 13)
        q = rb[..., 3:7]
        x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
        R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                      np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                      np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], -2)
        assert np.allclose(R[0, 0], quat_to_mat(q[0, 0]), atol=1e-12)
        rc = np.einsum("nlij,lj->nli", R, com)
        pc = rb[..., 0:3] + rc                                      # link centres of mass
        vc = rb[..., 7:10] + np.cross(rb[..., 10:13], rc)
        p = (mass[None, :, None] * vc).sum(1)
        cm = (mass[None, :, None] * pc).sum(1) / M
        Iw = np.einsum("nlij,ljk,nlmk->nlim", R, inertia, R)
        L = (np.cross(pc - cm[:, None], mass[None, :, None] * vc) + np.einsum("nlij,nlj->nli", Iw, rb[..., 10:13])).sum(1)
        return p, L

    g.set_reset(np.zeros(n, np.int64))
    g.simulate()                                     # one step to leave the reset state behind (targets = default)
    p0, L0 = momenta()
    rng = np.random.default_rng(3)
    steps = 20
    qd_max = np.zeros(n)
    for t in range(steps):
        g.pre_physics(rng.uniform(-0.08, 0.08, (n, 18)).astype(np.float32))
        g.simulate()
        qd_max = np.maximum(qd_max, np.abs(g.dof_state.reshape(n, 18, 2)[:, :, 1]).max(1))
    p1, L1 = momenta()
    free = qd_max < 6.0                       # envs in which no joint came near the 2 pi rad/s limit
    assert free.mean() > 0.97, free.mean()
    t_s = steps * float(cfg.dt)
    want = np.array([0.0, 0.0, -9.81 * M * t_s])
    # fp32 state, 40 substeps: the momentum of a 2.83 kg robot falling at 3.3 m/s is 9.3 N s; measured error: median 1e-3 per step
    # Round 6: a third of the reset draws start with the leg capsules overlapping (+-0.15 rad on the hip rolls) and the -- now stiff --
    # leg<->leg contact throws the legs apart in the first steps: the integrator's O(h) drift of those envs is larger (the fp64 oracle
    # shows the same numbers, 0.12 at 2 substeps -> 0.003 at 8 -> 0.0008 at 32: it is the integrator's, tests/test_oracle_round6.py)
    ep = np.abs((p1 - p0)[free] - want).max(1)
    dL = np.abs((L1 - L0)[free]).max()
    assert np.quantile(ep, 0.99) < 2e-2 and ep.max() < 0.3, (np.quantile(ep, 0.99), ep.max())
    assert dL < 2e-2, dL
