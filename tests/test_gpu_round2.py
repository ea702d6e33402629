"""GPU, round 2: the holes the round-1 review named -- the rigid-body refresh kernel against the oracle, the production
(wave-specialised) kernel's own post-physics pinned against the lane kernel on the same state, partial workgroups,
domain randomisation and PPO at the BASELINE sizes, checkpoint resume, and the reference's shipped policy played in
the HIP simulator (numeric fixture tests/golden/bez_kick_33_policy.npz)."""
import os

import numpy as np
import pytest
import torch

from bez_isaacgym_amd import abi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pair(n, **kw):
    from oracle.bez_oracle import Oracle
    from tests.sim_adapter import SimAdapter
    return Oracle(abi.default_config(n, **kw)), SimAdapter(abi.default_config(n, **kw))


def _sync(o, g):
    g.set_root_states(o.root_states); g.set_dof_state(o.dof_state); g.set_contact_forces(o.contact_forces)
    g.set_targets(o.targets); g.set_reset(o.reset_buf); g.set_progress(o.progress_buf)


def test_rigid_body_refresh_matches_oracle():
    """gym.refresh_rigid_body_state_tensor (kick_env.py:145,752): forward kinematics of the 21 robot bodies + ball row after
    random-action steps.  Positions 2e-4, quaternions up to sign 2e-4, velocities 2e-2 (same bars as the generalized state)."""
    n = 256
    o, g = _pair(n, seed=17)
    rng = np.random.default_rng(4)
    for t in range(6):
        _sync(o, g)
        act = rng.uniform(-1, 1, (n, 18)).astype(np.float32)
        o.step(act); g.step(act)
        _sync(o, g)  # identical generalized state -> the comparison isolates the refresh kernel
        ro, rg = o.rigid_body_states.reshape(n, 22, 13), g.rigid_body_states.reshape(n, 22, 13)
        np.testing.assert_allclose(rg[..., 0:3], ro[..., 0:3], atol=2e-4)
        sign = np.sign(np.sum(rg[..., 3:7] * ro[..., 3:7], axis=-1, keepdims=True))
        np.testing.assert_allclose(rg[..., 3:7] * sign, ro[..., 3:7], atol=2e-4)
        np.testing.assert_allclose(rg[..., 7:13], ro[..., 7:13], atol=2e-2)
    # the views KickEnv builds on it (kick_env.py:175-177): IMU link = body 1 rides on the torso origin
    np.testing.assert_allclose(rg[:, 1, 0:7], g.root_states.reshape(n, 2, 13)[:, 0, 0:7], atol=1e-6)


@pytest.mark.parametrize("task", ["bez_kick", "bez_walk", "bez_orient"])
@pytest.mark.parametrize("asset", ["default", "cleats", "box"])
def test_ws_kernel_post_physics_pinned_against_lane_kernel(task, asset):
    """The golden vectors run through the lane kernel's PRE/POST entry points; the production kernel has its own POST
    (per-role obs slots, partial pose-error sums, root bookkeeping, the task select of walk_env.py:826-1050 /
    orient_env.py:719-735,843-1018, the cleat flags of kick_env.py:1044-1069).  After a fused step, recompute observations and
    reward with the LANE kernel's obs-only pass on the very same resulting state -- for every task x asset: obs[0:36] must equal
    the dof state exactly, the remaining slots and the reward to 1e-6 of their scale, reset flags and feet flags exactly."""
    from tests.sim_adapter import SimAdapter
    from tests.test_tasks import make_cfg
    n = 448
    kw = dict(task=task, cleats=(asset == "cleats"), box=(asset == "box"), seed=23)
    a, b = SimAdapter(make_cfg(n, **kw)), SimAdapter(make_cfg(n, **kw))
    nobs = a.nobs
    rng = np.random.default_rng(12)
    a.set_obs_calls(1)  # past the process's first compute_imu call (quirk Q1): prev_lin_vel aliases the live velocity
    resets = 0
    for t in range(30):
        act = rng.uniform(-1, 1, (n, 18)).astype(np.float32)
        a.step(act)
        ds = a.dof_state.reshape(n, 18, 2)
        obs = a.obs
        np.testing.assert_array_equal(obs[:, 0:18], ds[:, :, 0])
        np.testing.assert_array_equal(obs[:, 18:36], ds[:, :, 1])
        b.set_root_states(a.root_states); b.set_dof_state(a.dof_state); b.set_contact_forces(a.contact_forces)
        if task != "bez_kick": b.set_goal(a.goal)
        b.set_progress(a.progress_buf); b.set_reset(np.zeros(n, np.int64)); b.set_obs_calls(1)
        b.observe_reward()
        np.testing.assert_allclose(b.obs[:, 36:nobs], obs[:, 36:nobs], atol=1e-6)
        np.testing.assert_array_equal(b.obs[:, 0:36], obs[:, 0:36])
        rew_a, rew_b = a.rew, b.rew
        np.testing.assert_allclose(rew_b, rew_a, atol=1e-6, rtol=1e-6)   # relative to its scale: walk / orient terminal rewards reach 1000
        np.testing.assert_array_equal(b.reset_buf, a.reset_buf)
        np.testing.assert_array_equal(b.feet, a.feet)
        resets += int(a.reset_buf.sum())
    assert resets + (a.progress_buf < 30).sum() > 0  # resets really happened inside the window


@pytest.mark.parametrize("other", ["lane", "ws8q"])
@pytest.mark.parametrize("variant", ["kick", "kick_cleats", "walk", "kick_box", "orient", "orient_cleats", "walk_box"])
def test_fused_step_kernels_agree(other, variant, monkeypatch):
    """The implementations of the fused control step -- 8 role waves (default), the same with four lanes per env (BEZ_SIM_KERNEL=ws8q,
    16-env workgroups) and one env per lane (BEZ_SIM_KERNEL=lane) -- compute the same physics in a different order.  From an identical state, with the same actions, resynchronised every step:
    integers exact, fp32 quantities to a few ulp of their scale.  N is not a multiple of the 64-env workgroup."""
    from tests.sim_adapter import SimAdapter
    from tests.test_tasks import make_cfg
    n = 200
    task = "bez_walk" if variant.startswith("walk") else ("bez_orient" if variant.startswith("orient") else "bez_kick")
    kw = dict(seed=31, task=task, cleats=variant.endswith("_cleats"), box=variant.endswith("_box"))
    monkeypatch.setenv("BEZ_SIM_KERNEL", "ws8")
    a = SimAdapter(make_cfg(n, **kw))
    monkeypatch.setenv("BEZ_SIM_KERNEL", other)  # read once, at bez_sim_create
    b = SimAdapter(make_cfg(n, **kw))
    monkeypatch.delenv("BEZ_SIM_KERNEL", raising=False)
    if variant.endswith("_box"):   # the box asset differs in the upper-body contact points: put a third of the envs on their back
        from tests.test_tasks import _lie_on_back
        _lie_on_back(a, n, n // 3)
    rng = np.random.default_rng(8)
    nres = 0
    from tests.parity_util import EnvOutliers
    T = EnvOutliers(n)
    for t in range(40):
        b.set_root_states(a.root_states); b.set_dof_state(a.dof_state); b.set_contact_forces(a.contact_forces)
        b.set_targets(a.targets); b.set_reset(a.reset_buf); b.set_progress(a.progress_buf); b.set_prev_lin_vel(a.prev_lin_vel)
        if task != "bez_kick": b.set_goal(a.goal)
        act = rng.uniform(-1, 1, (n, 18)).astype(np.float32)
        a.step(act); b.step(act)
        nres += int(a.reset_buf.sum())
        np.testing.assert_array_equal(b.progress_buf, a.progress_buf)
        np.testing.assert_array_equal(b.timeout_buf, a.timeout_buf)
        ra, rb = a.root_states, b.root_states
        # tolerances = 3x the worst difference seen over the 6 combinations (tools/kernel_ab_probe.py); all envs but a counted
        # handful per test (tests/parity_util.py: a speed-limit / saturation / contact switch within rounding of its boundary)
        T.close(rb[..., 0:7], ra[..., 0:7], 2e-5, what="pose")
        T.close(rb[..., 7:13], ra[..., 7:13], 4e-3, what="vel")
        da, db = a.dof_state.reshape(n, 18, 2), b.dof_state.reshape(n, 18, 2)
        T.close(db[..., 0], da[..., 0], 1e-4, what="q")
        T.close(db[..., 1], da[..., 1], 1.5e-2, what="qd")
        T.close(b.contact_forces, a.contact_forces, 2.5e-2, rtol=4e-3, what="cf")
        np.testing.assert_array_equal(b.obs[:, :36], np.concatenate([db[..., 0], db[..., 1]], axis=1))
        T.close(b.obs[:, 36:42], a.obs[:, 36:42], 3e-3, what="imu")
        qx, qy, qz, qw = (ra.reshape(n, -1, 13)[:, 0, 3 + k] for k in range(4))
        heading = (2 * (qw * qz + qx * qy)) ** 2 + (qw * qw + qx * qx - qy * qy - qz * qz) ** 2   # |(sin yaw, cos yaw)|^2 before normalisation
        ok = heading > 0.05   # a torso pointing straight up or down has no heading: the two slots amplify rounding without bound
        T.close(b.obs[:, 42:44], a.obs[:, 42:44], 2e-5, what="orn", rows=ok)
        assert np.mean(b.obs[:, 44:52] == a.obs[:, 44:52]) > 0.995  # threshold flags: a force within an ulp of 1 N / 0.01 N may flip
        bad = T.close(b.rew, a.rew, 4e-4 if task == "bez_walk" else 2e-5, what="rew") | T._bad  # walk rewards reach 1000
        np.testing.assert_array_equal(b.reset_buf[~bad], a.reset_buf[~bad])
        T.end_step()
    T.finish()
    assert nres > 0


def test_partial_workgroup_parity():
    """N = 100 is not a multiple of the 64 envs a workgroup owns: the masked tail lanes must not disturb the others."""
    n = 100
    o, g = _pair(n, seed=31)
    rng = np.random.default_rng(6)
    from tests.parity_util import EnvOutliers
    T = EnvOutliers(n)
    for t in range(15):
        _sync(o, g)
        act = rng.uniform(-1, 1, (n, 18)).astype(np.float32)
        o.step(act); g.step(act)
        T.close(g.dof_state.reshape(n, 18, 2)[..., 0], o.dof_state.reshape(n, 18, 2)[..., 0], 2e-4, what="q")
        T.close(g.obs[:, 36:44], o.obs[:, 36:44], 2e-2, what="imu")
        bad = T.close(g.contact_forces, o.contact_forces, 0.05, rtol=0.02, what="cf") | T._bad
        np.testing.assert_array_equal(g.reset_buf[~bad], o.reset_buf[~bad])
        T.end_step()
    T.finish()


def test_self_collision_and_ball_torso_contact_are_exercised():
    """The leg<->leg capsule contact and the ball<->torso box are part of the step: with them on, HIP and oracle agree
    (parity tests); here: switching them off changes the outcome, i.e. the tests above really cover them."""
    from tests.sim_adapter import SimAdapter
    n = 256
    on = SimAdapter(abi.default_config(n, seed=5))
    c = abi.default_config(n, seed=5); c.flags |= abi.FLAG_NO_SELF_COLLISION
    off = SimAdapter(c)
    rng = np.random.default_rng(1)
    for t in range(40):
        act = rng.uniform(-1, 1, (n, 18)).astype(np.float32)
        on.step(act); off.step(act)
    assert np.abs(on.dof_state - off.dof_state).max() > 1e-2
    # ball dropped onto the torso of a lying robot: the torso row reports the contact
    o, g = _pair(64, seed=2)
    rs = o.root_states.reshape(64, 2, 13).copy()
    rs[:, 0, 0:3] = [0.0, 0.0, 0.12]; rs[:, 0, 3:7] = [0.0, 0.70710678, 0.0, 0.70710678]  # pitched 90 deg: lying on its front
    rs[:, 1, 0:3] = rs[:, 0, 0:3] + [0.03, 0.0, 0.11]; rs[:, 1, 7:10] = [0, 0, -0.5]
    o.set_root_states(rs.reshape(-1, 13)); g.set_root_states(rs.reshape(-1, 13))
    g.set_dof_state(o.dof_state)
    hit = 0
    for t in range(6):
        _sync(o, g)
        act = np.zeros((64, 18), np.float32)
        o.simulate(); g.simulate()
        cf_o, cf_g = o.contact_forces.reshape(64, 22, 3), g.contact_forces.reshape(64, 22, 3)
        np.testing.assert_allclose(cf_g[:, [0, 21]], cf_o[:, [0, 21]], rtol=0.03, atol=0.1)
        hit += int((np.linalg.norm(cf_o[:, 21], axis=1) > 0.5).sum())
    assert hit > 0


def test_domain_randomization_parity_full_size():
    """BASELINE config 5's per-GPU workload (4096 envs): every per-env array -- friction, Kp, Kd, mass, gravity and the
    jittered physical joint limits (bez_kick.yaml:151-219) -- changes the HIP step exactly as it changes the oracle's."""
    n = 4096
    o, g = _pair(n, seed=3)
    rng = np.random.default_rng(8)
    import json
    model = json.load(open(os.path.join(ROOT, "bez_isaacgym_amd", "model", "bez_model.json")))
    lo, hi = np.array(model["dof_lower"], np.float32), np.array(model["dof_upper"], np.float32)
    params = {abi.PARAM_FRICTION: rng.uniform(0.7, 1.3, (n, 1)), abi.PARAM_KP_SCALE: rng.uniform(0.5, 1.5, (n, 18)),
              abi.PARAM_KD_SCALE: rng.uniform(0.5, 1.5, (n, 18)), abi.PARAM_MASS_SCALE: rng.uniform(0.5, 1.5, (n, 19)),
              abi.PARAM_GRAVITY: np.tile([[0.0, 0.0, -9.81]], (n, 1)) + rng.normal(0, 0.3, (n, 3)),
              abi.PARAM_DOF_LOWER: lo[None] + rng.normal(0, 0.2, (n, 18)), abi.PARAM_DOF_UPPER: hi[None] + rng.normal(0, 0.2, (n, 18))}
    for k, v in params.items():
        o.set_env_params(k, v.astype(np.float32)); g.set_env_params(k, v.astype(np.float32))
    from tests.parity_util import EnvOutliers
    T = EnvOutliers(n)
    for t in range(4):
        _sync(o, g)
        act = rng.uniform(-1, 1, (n, 18)).astype(np.float32)
        o.step(act); g.step(act)
        do, dg = o.dof_state.reshape(n, 18, 2), g.dof_state.reshape(n, 18, 2)
        T.close(dg[..., 0], do[..., 0], 1.5e-4, what="q")
        T.close(dg[..., 1], do[..., 1], 1.5e-2, what="qd")
        ro, rg = o.root_states.reshape(n, 2, 13), g.root_states.reshape(n, 2, 13)
        bad = T.close(rg[..., 0:7], ro[..., 0:7], 2e-4, what="pose") | T._bad
        np.testing.assert_array_equal(g.reset_buf[~bad], o.reset_buf[~bad])
        T.end_step()
    T.finish()
    # the limit jitter alone is visible: same run with model limits differs
    base_o, _ = _pair(256, seed=3)
    lim_o, _ = _pair(256, seed=3)
    lim_o.set_env_params(abi.PARAM_DOF_UPPER, (hi[None] - 0.6 + 0 * rng.normal(0, 1, (256, 18))).astype(np.float32))
    for t in range(10):
        act = rng.uniform(-1, 1, (256, 18)).astype(np.float32)
        base_o.step(act); lim_o.step(act)
    assert np.abs(base_o.dof_state - lim_o.dof_state).max() > 1e-2


def _agent(num_envs, minibatch, randomize=False, **cfg_over):
    from bez_isaacgym_amd.ppo.a2c_continuous import A2CAgent
    from bez_isaacgym_amd.utils.config import load_config
    from bez_isaacgym_amd.utils.rlgames_utils import RLGPUEnv, get_rlgames_env_creator
    args = ["task=bez_kick", "num_envs=%d" % num_envs, "headless=True"] + (["task.task.randomize=True"] if randomize else [])
    cfg = load_config(args)
    cfg["task"]["seed"] = 42
    creator = get_rlgames_env_creator(cfg["task"], "bez_kick", "cuda:0", "cuda:0", 0, True)
    venv = RLGPUEnv("rlgpu", num_envs, env_creator=creator)
    params = cfg["train"]["params"]
    params["config"].update(minibatch_size=minibatch, save_frequency=0, save_best_after=10 ** 9, **cfg_over)
    return A2CAgent(params, venv, "cuda:0")


def test_ppo_config3_full_size():
    """BASELINE config 3 as written: num_envs 4096, horizon 32, minibatch 32768, 5 mini-epochs, AMP; eager warm-up epochs,
    then the HIP-graph capture and replays."""
    agent = _agent(4096, 32768)
    assert agent.batch_size == 131072 and agent.num_minibatches == 4 and agent.mini_epochs == 5 and agent.mixed_precision
    agent.obs = agent.env_reset()
    stats = [agent.train_epoch() for _ in range(5)]
    assert agent.use_graphs and agent._g_rollout is not None and agent._g_update is not None
    assert all(np.isfinite([s["kl"], s["a_loss"], s["c_loss"]]).all() for s in stats)
    assert agent.frame == 5 * 131072 and len(agent.game_rewards) > 0
    assert all(torch.isfinite(p).all() for p in agent.model.parameters())


def test_ppo_with_domain_randomization_is_graph_captured():
    """with task.randomize=True the per-env redraws happen inside the simulator (no host sync in
    step()), so the env is graph-safe and the PPO loop captures and replays its epochs exactly as without DR."""
    agent = _agent(512, 4096, randomize=True)
    assert agent.use_graphs
    agent.obs = agent.env_reset()
    stats = [agent.train_epoch() for _ in range(5)]
    assert agent._g_rollout is not None and agent._g_update is not None     # captured, and the later epochs were replays
    assert all(np.isfinite([s["kl"], s["a_loss"], s["c_loss"]]).all() for s in stats)
    env = agent.vec_env.env
    assert env.sim.get_env_params(abi.PARAM_DOF_LOWER).shape == (512, 18) and torch.isfinite(env.sim.get_env_params(abi.PARAM_FRICTION)).all()
    assert int(env.randomize_buf.max()) == 5 * 32 + 1    # counted on the device through eager, capturing and replayed epochs (+ env_reset's step)


def test_resume_keeps_adaptive_lr_connected(tmp_path):
    """Optimizer.load_state_dict replaces the lr entries of the param groups: after restore() they must again BE the device
    tensor the adaptive-KL rule writes, carry the saved value, and the first epochs after a resume run eagerly."""
    agent = _agent(256, 2048)
    agent.obs = agent.env_reset()
    for _ in range(3):
        agent.train_epoch()
    agent.lr_t.fill_(1.234e-4)
    path = str(tmp_path / "ck.pth")
    agent.save(path)
    other = _agent(256, 2048)
    other.restore(path)
    assert other.optimizer.param_groups[0]["lr"] is other.lr_t
    assert abs(float(other.lr_t) - 1.234e-4) < 1e-9 and other.epoch_num == 3
    other.obs = other.env_reset()
    other.train_epoch()
    assert other._g_rollout is None  # eager warm-up first, whatever epoch_num says
    lr_before = float(other.lr_t)
    other.scheduler.update_(other.lr_t, torch.tensor(1.0, device="cuda:0"))  # huge KL -> lr must drop, in the optimizer too
    assert float(other.optimizer.param_groups[0]["lr"]) < lr_before


def test_player_loads_reference_policy_fixture():
    """The numeric fixture of the reference's shipped policy loads into the player (utils/players.py:68-72 restore)."""
    from bez_isaacgym_amd.utils.player import PpoPlayerContinuous
    p = PpoPlayerContinuous(os.path.join(ROOT, "tests", "golden", "bez_kick_33_policy.npz"), "cuda:0")
    assert abs(float(p.running_mean_std.running_mean[38]) - 0.98) < 0.02 and p.checkpoint["epoch"] == 6156
    a = p.get_action(torch.zeros(4, 54, device="cuda:0"))
    assert a.shape == (4, 18) and float(a.abs().max()) <= 1.0


_S2S = {}


def _reference_policy_rollout():
    if not _S2S:
        import sys
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from sim2sim_gpu import evaluate
        from bez_isaacgym_amd.utils.player import PpoPlayerContinuous
        player = PpoPlayerContinuous(os.path.join(ROOT, "tests", "golden", "bez_kick_33_policy.npz"), "cuda:0")
        _S2S.update(evaluate(player, None, n=4096, steps=900, seed=1))
        print("reference policy in the HIP sim:", {k: v for k, v in _S2S.items() if k not in ("obs_z", "obs_std_ratio")})
    return _S2S


@pytest.mark.xfail(strict=True, reason="DESIGN.md 6.1: the un-pinned physics does not reproduce PhysX for this policy (1.7 % goals, return -1.0; "
                   "PhysX 87.55).  The bars below are the ACCEPTANCE level of the round-2 review; strict: reaching them must be noticed")
def test_reference_policy_reaches_physx_level():
    """The acceptance test of the physics: the reference's shipped policy, which scores 87.55 under PhysX,
    played in 4096 HIP envs: goal rate >= 0.5, mean return >= 40, observation statistics within 2 sigma of the checkpoint's."""
    r = _reference_policy_rollout()
    assert r["goal_rate"] >= 0.5 and r["mean_return"] >= 40.0 and np.abs(np.asarray(r["obs_z"])[:52]).max() < 2.0


def test_reference_policy_in_hip_sim():
    """the reference's shipped policy (Bez_Kick_33, 87.55 mean reward under PhysX) played deterministically
    (utils/players.py:46-66) in 4096 HIP envs for a whole 900-step horizon.  This is the only reference-held evidence that
    can speak about the un-pinned physics.  MEASURED outcome: it kicks the ball (about 1 m/s after 5
    control steps, as under PhysX) but loses its balance about 50 steps later, before the ball reaches the goal.  These
    assertions are only a REGRESSION FLOOR of what the build reaches today (the acceptance bars are the xfail test above)."""
    r = _reference_policy_rollout()
    assert r["episodes"] > 4096
    assert 35.0 < r["mean_length"] < 900.0          # PhysX: ~112 steps to the goal; here ~50 until it falls
    assert r["goal_rate"] >= 0.01                    # PhysX: essentially always
    assert r["mean_return"] > -3.0                   # PhysX: 87.55
    # the kick itself transfers: goals, when they happen, take about as long as under PhysX (100 - 100*t/900 = 87.55 -> t = 112)
    assert 50.0 < r["goal_length"] < 200.0
    z = np.asarray(r["obs_z"])
    assert np.abs(z[0:36]).max() < 3.0               # joint-space statistics stay within 3 sigma of the checkpoint's
    assert abs(z[52]) < 1e-3 and abs(z[53]) < 1e-3   # constant ball_init tail (Q5)


def test_play_mode_runs_the_reference_checkpoint_fixture():
    """train.py test=True checkpoint=... (reference play.py / README): whole episodes with the deterministic player."""
    from bez_isaacgym_amd.train import launch
    mean_r, mean_s, played = launch(["task=bez_kick", "num_envs=512", "headless=True", "test=True",
                                     "checkpoint=" + os.path.join(ROOT, "tests", "golden", "bez_kick_33_policy.npz"),
                                     "train.params.config.player={games_num: 600, max_steps: 400}"])
    assert played >= 600 and 20 < mean_s < 900 and np.isfinite(mean_r)
