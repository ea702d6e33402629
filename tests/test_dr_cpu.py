"""CPU: the oracle's restatement of VecTask.apply_randomizations (vec_task.py:505-725) as the simulator runs it on the device:
clocks (randomize_buf, frequency, frame-keyed gravity), the linear schedule, the 500 friction buckets, config translation."""
import numpy as np
import pytest

from bez_isaacgym_amd import abi
from oracle.bez_oracle import Oracle


def _params(sched=3000, freq=600):
    lin = {"schedule": "linear", "schedule_steps": sched} if sched else {}
    return {"frequency": freq,
            "observations": {"range": [0, .002], "operation": "additive", "distribution": "gaussian"},
            "actions": {"range": [0., .02], "operation": "additive", "distribution": "gaussian"},
            "sim_params": {"gravity": dict(range=[0, 0.4], operation="additive", distribution="gaussian", **lin)},
            "actor_params": {"bez": {"color": True,
                "rigid_body_properties": {"mass": dict(range=[0.5, 1.5], operation="scaling", distribution="uniform", setup_only=True, **lin)},
                "rigid_shape_properties": {"friction": dict(num_buckets=500, range=[0.7, 1.3], operation="scaling", distribution="uniform", **lin),
                                           "restitution": dict(range=[0., 0.7], operation="scaling", distribution="uniform", **lin)},
                "dof_properties": {"damping": dict(range=[0.5, 1.5], operation="scaling", distribution="uniform", **lin),
                                   "stiffness": dict(range=[0.5, 1.5], operation="scaling", distribution="uniform", **lin),
                                   "lower": dict(range=[0, 0.01], operation="additive", distribution="gaussian", **lin),
                                   "upper": dict(range=[0, 0.01], operation="additive", distribution="gaussian", **lin)}}}}


def test_dr_config_translation_matches_the_task_yaml():
    from bez_isaacgym_amd.utils.config import load_config
    cfg = load_config(["task=bez_kick", "num_envs=64", "headless=True"])
    d = abi.dr_config_from_params(cfg["task"]["task"]["randomization_params"])   # bez_kick.yaml:151-219
    assert d.frequency == 600 and d.friction_buckets == 500
    assert (d.friction.a, d.friction.b, d.friction.enabled, d.friction.schedule_steps) == (np.float32(0.7), np.float32(1.3), 1, 3000)
    assert (d.stiffness.a, d.stiffness.b, d.damping.a, d.damping.b) == (0.5, 1.5, 0.5, 1.5)
    assert (d.lower.b, d.upper.b, d.gravity.b) == (np.float32(0.01), np.float32(0.01), np.float32(0.4)) and d.gravity.schedule_steps == 3000
    assert d.observations.enabled and d.observations.schedule_steps == 0 and d.actions.b == np.float32(0.02)
    with pytest.raises(ValueError):
        abi.dr_config_from_params({"observations": {"range": [0, 1], "operation": "scaling", "distribution": "gaussian"}})


def test_dr_clocks_schedule_and_buckets():
    n = 48
    o = Oracle(abi.default_config(n, seed=3))
    o.set_randomization(abi.dr_config_from_params(_params(sched=20, freq=4)))
    # frame 0 of a linear schedule: first_randomization leaves everything nominal (vec_task.py:560-566)
    np.testing.assert_array_equal(o.get_env_params(abi.PARAM_FRICTION), 1.0)
    np.testing.assert_array_equal(o.get_env_params(abi.PARAM_KP_SCALE), 1.0)
    np.testing.assert_array_equal(o.get_env_params(abi.PARAM_GRAVITY), np.tile(np.float32([0, 0, -9.81]), (n, 1)))
    np.testing.assert_array_equal(o.dr_noise, np.float32([0, 0.002, 0, 0.02]))   # no schedule on the noise entries: full size at once
    act = np.zeros((n, 18), np.float32)
    # nobody resets: randomize_buf counts, nothing is redrawn, gravity untouched (apply_randomizations is only reached from reset_idx)
    for t in range(6):
        o.step(act)
    np.testing.assert_array_equal(o.randomize_buf, 6)
    np.testing.assert_array_equal(o.get_env_params(abi.PARAM_KP_SCALE), 1.0)
    np.testing.assert_array_equal(o.get_env_params(abi.PARAM_GRAVITY)[:, 2], np.float32(-9.81))
    # flag envs 0..9: they are past `frequency`, so they redraw at frame 7 (schedule 7 / 20) and restart their clock; the others keep counting
    rst = np.zeros(n, np.int64); rst[:10] = 1
    o.set_reset(rst)
    o.step(act)
    rb = o.randomize_buf
    np.testing.assert_array_equal(rb[:10], 0); np.testing.assert_array_equal(rb[10:], 7)
    kp = o.get_env_params(abi.PARAM_KP_SCALE)
    np.testing.assert_array_equal(kp[10:], 1.0)
    s = np.float32(7) / np.float32(20)
    assert np.all(kp[:10] >= 1 - 0.5 * s - 1e-6) and np.all(kp[:10] <= 1 + 0.5 * s + 1e-6) and kp[:10].std() > 0.05
    g = o.get_env_params(abi.PARAM_GRAVITY)
    assert np.all(g == g[0]) and np.any(g[0] != np.float32([0, 0, -9.81]))    # one draw for the whole sim, 7 frames after frame 0 >= frequency
    # an env that resets again before `frequency` steps keeps its parameters
    rst[:] = 0; rst[0] = 1
    o.set_reset(rst); o.step(act)
    np.testing.assert_array_equal(o.get_env_params(abi.PARAM_KP_SCALE)[0], kp[0])
    assert o.randomize_buf[0] == 1
    # schedule complete: full ranges, 500 friction buckets, limits jittered by N(0, 0.01)
    for t in range(30):
        o.set_reset(np.ones(n, np.int64)); o.step(act)
    f = o.get_env_params(abi.PARAM_FRICTION)[:, 0]
    k = (f - 0.7) / 0.6 * 499
    np.testing.assert_allclose(k, np.round(k), atol=2e-3)
    assert f.min() < 0.85 and f.max() > 1.15
    lo = o.get_env_params(abi.PARAM_DOF_LOWER) - np.float32(o.get_env_params(abi.PARAM_DOF_LOWER).mean(axis=0, keepdims=True))
    assert 0.006 < lo.std() < 0.014


def test_dr_redraw_is_keyed_by_global_env_and_episode():
    dr = abi.dr_config_from_params(_params(sched=0, freq=1))
    a = Oracle(abi.default_config(32, seed=8)); a.set_randomization(dr)
    b = Oracle(abi.default_config(16, seed=8, env_id_offset=16)); b.set_randomization(dr)
    for p in (abi.PARAM_FRICTION, abi.PARAM_KD_SCALE, abi.PARAM_DOF_LOWER):
        np.testing.assert_array_equal(a.get_env_params(p)[16:], b.get_env_params(p))
    c = Oracle(abi.default_config(32, seed=9)); c.set_randomization(dr)
    assert np.abs(c.get_env_params(abi.PARAM_KD_SCALE) - a.get_env_params(abi.PARAM_KD_SCALE)).max() > 0.1   # another seed, other draws
