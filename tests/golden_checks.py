"""Golden-vector checks shared by the oracle tests (CPU) and the HIP parity tests (GPU).

`backend` is any object with the small state-injection interface both the oracle wrapper
(oracle/bez_oracle.py: Oracle) and the HIP sim test adapter (tests/sim_adapter.py) provide:
set_root_states / set_dof_state / set_contact_forces / set_prev_lin_vel / set_reset / set_progress /
set_flags / set_obs_calls / pre_physics / post_physics / observe_reward and the obs / rew / reset_buf /
progress_buf / timeout_buf / targets / contact_forces / feet getters.
Golden arrays come from the reference's own code (tests/golden/make_golden.py).
"""
import numpy as np

N = 64
FLAG_ALIAS = 1
# fp32 tolerance for everything the reference owns (<= 1e-5 abs)
ATOL = 1e-5


def _inject(b, root_pos=None, quat=None, vel=None, ang=None, dof_pos=None, dof_vel=None, ball=None, ball_v=None,
            cf=None, prev=None, reset=None, progress=None):
    n = N
    root = np.zeros((n, 2, 13), np.float32)
    root[:, 0, 2] = 0.32
    root[:, :, 6] = 1.0
    root[:, 1, 0:3] = [0.3, 0.0, 0.08]
    if root_pos is not None: root[:, 0, 0:3] = root_pos
    if quat is not None: root[:, 0, 3:7] = quat
    if vel is not None: root[:, 0, 7:10] = vel
    if ang is not None: root[:, 0, 10:13] = ang
    if ball is not None: root[:, 1, 0:3] = ball
    if ball_v is not None: root[:, 1, 7:10] = ball_v
    b.set_root_states(root.reshape(n * 2, 13))
    dof = np.zeros((n, 18, 2), np.float32)
    if dof_pos is not None: dof[:, :, 0] = dof_pos
    if dof_vel is not None: dof[:, :, 1] = dof_vel
    b.set_dof_state(dof.reshape(n * 18, 2))
    c = np.zeros((n, 22, 3), np.float32)
    if cf is not None: c[:] = cf
    b.set_contact_forces(c.reshape(n * 22, 3))
    b.set_prev_lin_vel(np.zeros((n, 3), np.float32) if prev is None else prev)
    b.set_reset(np.zeros(n, np.int64) if reset is None else reset)
    b.set_progress(np.zeros(n, np.int64) if progress is None else progress)


def check_imu(b, G):
    """compute_imu + quaternion_to_matrix (kick_env.py:857-930): explicit prev, first call, aliased prev."""
    b.set_flags(0)  # true finite difference against the stored prev
    _inject(b, quat=G["imu_quat"], vel=G["imu_vel"], ang=G["imu_ang"], prev=G["imu_prev"])
    b.observe_reward()
    np.testing.assert_allclose(b.obs[:, 36:42], G["imu_out"], atol=2e-4, rtol=1e-5)  # (v-prev)/dt amplifies fp32 ulp by 60
    np.testing.assert_allclose(b.prev_lin_vel, G["imu_newprev"], atol=0)
    _inject(b, quat=G["imu_quat"], vel=G["imu_vel"], ang=G["imu_ang"])  # prev = zeros: first call of a process
    b.observe_reward()
    np.testing.assert_allclose(b.obs[:, 36:42], G["imu_out_first"], atol=2e-4, rtol=1e-5)
    b.set_flags(FLAG_ALIAS)
    b.set_obs_calls(1)  # any call after the first: prev aliases the live tensor (quirk Q1)
    _inject(b, quat=G["imu_quat"], vel=G["imu_vel"], ang=G["imu_ang"], prev=G["imu_prev"])
    b.observe_reward()
    np.testing.assert_allclose(b.obs[:, 36:42], G["imu_out_alias"], atol=ATOL)


def check_off_orn(b, G):
    """compute_off_orn (kick_env.py:933-962)."""
    b.set_flags(FLAG_ALIAS); b.set_obs_calls(1)
    _inject(b, root_pos=G["orn_pos"], quat=G["orn_quat"])
    b.observe_reward()
    np.testing.assert_allclose(b.obs[:, 42:44], G["orn_out"], atol=ATOL)


def check_feet(b, G):
    """compute_feet_sensors_no_cleats (kick_env.py:966-1040) incl. the in-place noise filter."""
    b.set_flags(FLAG_ALIAS); b.set_obs_calls(1)
    cf = np.zeros((N, 22, 3), np.float32)
    cf[:, 12] = G["feet_in"]
    cf[:, 20] = G["feet_in"][::-1]
    _inject(b, cf=cf)
    b.observe_reward()
    np.testing.assert_array_equal(b.obs[:, 44:48], G["feet_out"])
    np.testing.assert_array_equal(b.obs[:, 48:52], G["feet_out"][::-1])
    np.testing.assert_array_equal(b.feet[:, 0:4], G["feet_out"])
    after = b.contact_forces.reshape(N, 22, 3)
    np.testing.assert_array_equal(after[:, 12], G["feet_filtered"])
    np.testing.assert_array_equal(after[:, 20], G["feet_filtered"][::-1])


def check_reward(b, G, tag):
    """compute_bez_reward (kick_env.py:1198-1395) and the obs layout (kick_env.py:1398-1417)."""
    b.set_flags(FLAG_ALIAS); b.set_obs_calls(1)
    g = lambda k: G["rew_%s_%s" % (tag, k)]
    _inject(b, root_pos=g("root"), quat=g("quat"), vel=g("v_imu"), ang=g("w_imu"), dof_pos=g("dof_pos"),
            dof_vel=g("dof_vel"), ball=g("ball"), ball_v=g("ball_v"), reset=g("reset"), progress=g("progress"))
    b.observe_reward()
    np.testing.assert_allclose(b.rew, g("rew"), atol=2e-5, rtol=1e-5)
    np.testing.assert_array_equal(b.reset_buf, g("rst"))
    obs = b.obs
    np.testing.assert_allclose(obs[:, 0:18], g("dof_pos"), atol=0)
    np.testing.assert_allclose(obs[:, 18:36], g("dof_vel"), atol=0)
    np.testing.assert_allclose(obs[:, 52:54], np.tile([[0.175, 0.0]], (N, 1)), atol=1e-7)


def check_pre_physics(b, G):
    """vec_task.py:317 clamp + KickEnv.pre_physics_step (kick_env.py:410-419)."""
    b.pre_physics(G["pre_actions"])
    np.testing.assert_allclose(b.targets, G["pre_targets"], atol=1e-7)


def check_step_sequence(b, G):
    """VecTask.step (vec_task.py:303-349) x6 on a scripted simulator: bookkeeping, obs, reward, reset flags."""
    b.set_flags(FLAG_ALIAS)
    b.set_obs_calls(0)
    S = int(G["seq_len"])
    b.set_progress(G["seq_progress0"])
    b.set_reset(np.zeros(N, np.int64))
    b.set_prev_lin_vel(np.zeros((N, 3), np.float32))
    for t in range(S):
        b.pre_physics(G["seq_actions"][t])
        np.testing.assert_allclose(b.targets, G["seq%d_targets" % t], atol=1e-7)
        # "gym.simulate": the scripted state appears in the sim tensors
        b.set_root_states(G["seq%d_root" % t]); b.set_dof_state(G["seq%d_dof" % t]); b.set_contact_forces(G["seq%d_cf" % t])
        b.post_physics()
        # step 0 takes (v - 0)/dt (amplified ulps); later steps see the aliased prev
        np.testing.assert_allclose(b.obs, G["seq%d_obs" % t], atol=2e-4 if t == 0 else ATOL, rtol=1e-5)
        np.testing.assert_allclose(b.rew, G["seq%d_rew" % t], atol=2e-5, rtol=1e-5)
        np.testing.assert_array_equal(b.reset_buf, G["seq%d_reset" % t])
        np.testing.assert_array_equal(b.timeout_buf, G["seq%d_timeout" % t])
        np.testing.assert_array_equal(b.progress_buf, G["seq%d_progress" % t])
        np.testing.assert_array_equal(b.contact_forces.reshape(N, 22, 3)[:, [12, 20]],
                                      G["seq%d_cf_after" % t].reshape(N, 22, 3)[:, [12, 20]])
        b.set_reset(np.zeros(N, np.int64))  # the generator keeps the sequence reset-free the same way


ALL_CHECKS = [check_imu, check_off_orn, check_feet, lambda b, G: check_reward(b, G, "normal"),
              lambda b, G: check_reward(b, G, "edge"), check_pre_physics, check_step_sequence]
