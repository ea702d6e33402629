"""Scripted scenarios on the CPU oracle (missing 3): the reference's get-up tables from lying starts and
the per-DOF limit sweep of `bez_isaacgym/test/test_kick_env.py:142-186`.  The same scenarios run on the HIP simulator in
tests/test_gpu_round5.py.  What the get-ups reach is MEASURED here, not assumed: nothing under /root/reference records that these
open-loop tables (written for the soccerbot's PyBullet model) succeed under Isaac Gym either."""
import numpy as np
import pytest

from bez_isaacgym_amd import abi
from tests.scenarios import dof_sweep, lay_down, make_backend, play


def _getup(name, n=8, flags=None):
    import json, os
    from tests.scenarios import ROOT
    model = json.load(open(os.path.join(ROOT, "bez_isaacgym_amd", "model", "bez_model.json")))
    cfg = abi.default_config(n, seed=7)
    if flags is not None:
        cfg.flags = flags
    sim = make_backend("oracle", cfg)
    sim.step(np.zeros((n, 18), np.float32))
    lay_down(sim, n, name, np.random.default_rng(3))
    return play(sim, n, name, model)


def test_getup_front_reaches_the_squat_oracle():
    """`simulation_getupfront` from lying face down (the yaml's own "flat" quaternion, bez_kick.yaml:20): the arms push the torso up and
    the legs fold under it -- the robot reaches the squat on its feet (torso 0.22 m high, 33 degrees forward) in every env.  Regression floor."""
    r = _getup("getupfront")
    assert r["finite"] == 1.0 and r["max_z"] > 0.20 and r["max_up"] > 0.75, r


def test_getup_side_rolls_onto_the_front_oracle():
    """`simulation_getupside` only swings the arms back (-pi/2): from its side the robot ends lying on its front, where getupfront starts."""
    r = _getup("getupside")
    assert r["finite"] == 1.0 and r["final_z"] < 0.12 and abs(r["final_up"]) < 0.4, r


@pytest.mark.xfail(strict=True, reason="measured: 0 of 8 envs stand at the end of simulation_getupfront (squat reached, max z 0.219 / up 0.80; in the "
                                       "last key frame the torso pitches forward over the toes, final z 0.077); identical at damping 2 and effort 5 N*m -- "
                                       "profiles/r05_getup.txt")
def test_getup_front_ends_standing_oracle():
    assert _getup("getupfront")["standing"] >= 0.9


@pytest.mark.xfail(strict=True, reason="measured: with the baked contact set (feet + 14 upper-body guard points) the robot never rolls off its back "
                                       "(max z 0.092); with ground contact at every collision shape's corners (BEZ_FLAG_ALL_GROUND_SHAPES, oracle only) "
                                       "it rolls over and reaches the same squat as getupfront, then tips forward the same way -- profiles/r05_getup.txt")
def test_getup_back_ends_standing_oracle():
    assert _getup("getupback")["standing"] >= 0.9


def test_all_ground_shapes_let_the_back_getup_roll_over_oracle():
    """Oracle-only variant: knees, hips and forearm boxes touch the ground as well.  The back get-up then rolls the robot over and reaches
    the squat (without them it stays on its back): what the get-ups need is contact geometry, not drive authority."""
    a = _getup("getupback", n=4)
    b = _getup("getupback", n=4, flags=abi.FLAG_IMU_PREV_ALIAS | abi.FLAG_ALL_GROUND_SHAPES)
    assert a["max_z"] < 0.12 and b["max_z"] > 0.20 and b["max_up"] > 0.75, (a, b)


def check_dof_sweep(rows):
    """Every actuated DOF reaches both limits and returns (zero gravity, floating base) -- except where the model says it cannot:
    the head joints never receive an action (kick_env.py:414), a hip rolling INWARD meets the other leg (collision_filter 0,
    kick_env.py:365-366: the foot plates, 8 mm apart in the default pose, touch first) 0.6 rad short of its -0.785 rad limit, and a hip
    yawing to its lower limit swings the hip box into the other leg 0.1 rad short of -1.309.  (Round 5's 3000 N/m explicit spring let
    the saturated drive push 2 cm into the other leg: its hip roll stopped 0.34 rad short, at the thighs.)"""
    for r in rows:
        assert r["finite"], r
        if r["dof"] < 2:
            assert r["miss_default"] < 1e-3 and r["miss_lower"] > 1.5, r       # did not move
        elif r["name"].endswith("leg_motor_1"):
            assert 0.5 < r["miss_lower"] < 0.72 and r["miss_upper"] < 0.05 and r["miss_default"] < 0.05, r
        elif r["name"].endswith("leg_motor_0"):
            assert 0.03 < r["miss_lower"] < 0.2 and r["miss_upper"] < 0.05 and r["miss_default"] < 0.05, r
        else:
            assert r["miss_lower"] < 0.05 and r["miss_upper"] < 0.05 and r["miss_default"] < 0.05, r


def test_dof_sweep_oracle(model):
    n = 2
    cfg = abi.default_config(n, seed=3)
    cfg.gravity[:] = [0.0, 0.0, 0.0]
    sim = make_backend("oracle", cfg)
    sim.step(np.zeros((n, 18), np.float32))
    check_dof_sweep(dof_sweep(sim, n, model))


def test_separating_axis_test_of_the_pair_log():
    """tools/pair_penetration.obb_separation on configurations with known answers: face-to-face gap and overlap, a rotated box whose corner
    dips into a face, an edge-edge case only a cross-product axis separates, and the degenerate parallel-axes case."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from pair_penetration import obb_separation, quat_to_mat
    I = np.eye(3)[None]
    h = np.array([0.5, 0.5, 0.5])
    sep = lambda ca, Ra, ha, cb, Rb, hb: float(obb_separation(np.array([ca], float), Ra, np.asarray(ha, float), np.array([cb], float), Rb, np.asarray(hb, float))[0])
    assert sep([0, 0, 0], I, h, [1.3, 0, 0], I, h) == pytest.approx(0.3)          # face to face, 0.3 apart (parallel axes: cross products degenerate)
    assert sep([0, 0, 0], I, h, [0.9, 0.2, 0.1], I, h) == pytest.approx(-0.1)     # overlapping by 0.1 along x (the least-penetration axis)
    # box B rotated 45 deg about z: its corner points at A's +x face; corner reach = 0.5 * sqrt(2)
    q = np.array([[0.0, 0.0, np.sin(np.pi / 8), np.cos(np.pi / 8)]])
    R45 = quat_to_mat(q)
    reach = 0.5 * np.sqrt(2.0)
    assert sep([0, 0, 0], I, h, [0.5 + reach + 0.05, 0, 0], R45, h) == pytest.approx(0.05, abs=1e-9)
    assert sep([0, 0, 0], I, h, [0.5 + reach - 0.02, 0, 0], R45, h) == pytest.approx(-0.02, abs=1e-9)
    # two long thin rods crossing at right angles, offset along z: only z separates them
    assert sep([0, 0, 0], I, [2.0, 0.05, 0.05], [0, 0, 0.3], I, [0.05, 2.0, 0.05]) == pytest.approx(0.2)
    # a rod along x and a rod along (y + z) / sqrt 2 whose centre sits 0.4 above: the lines are 0.4 / sqrt 2 apart along n = (0, -1, 1) / sqrt 2,
    # which is B's own thin axis; A's square cross-section reaches 0.05 sqrt 2 along n, B's 0.05
    qx = np.array([[np.sin(np.pi / 8), 0.0, 0.0, np.cos(np.pi / 8)]])      # 45 deg about x
    Rx = quat_to_mat(qx)
    d = sep([0, 0, 0], I, [1.0, 0.05, 0.05], [0, 0.0, 0.4], Rx, [0.05, 1.0, 0.05])
    assert d == pytest.approx(0.4 / np.sqrt(2.0) - 0.05 * np.sqrt(2.0) - 0.05, abs=1e-9)


def check_fixed_base_sweep(rows, root_before, root_after):
    """The reference's form of the sweep ("better when fixBaseLink = True", test/test_kick_env.py:142-186): gravity on, the torso welded one
    metre above the plane.  The torso does not move by a bit; every actuated DOF reaches both limits under its limb's weight and returns,
    with the same exceptions as the floating sweep (head joints never commanded, a hip rolling inward or yawing to its lower limit meets the other leg)."""
    assert np.array_equal(root_before, root_after)
    for r in rows:
        assert r["finite"], r
        if r["dof"] < 2:
            assert r["miss_default"] < 1e-3 and r["miss_lower"] > 1.5, r
        elif r["name"].endswith("leg_motor_1"):
            assert 0.5 < r["miss_lower"] < 0.72 and r["miss_upper"] < 0.06 and r["miss_default"] < 0.06, r
        elif r["name"].endswith("leg_motor_0"):
            assert 0.03 < r["miss_lower"] < 0.2 and r["miss_upper"] < 0.06 and r["miss_default"] < 0.06, r
        else:
            assert r["miss_lower"] < 0.06 and r["miss_upper"] < 0.06 and r["miss_default"] < 0.06, r


def test_dof_sweep_with_a_fixed_base_oracle(model):
    n = 2
    cfg = abi.default_config(n, seed=3)
    cfg.flags |= abi.FLAG_FIX_BASE
    sim = make_backend("oracle", cfg)
    sim.step(np.zeros((n, 18), np.float32))
    rows = dof_sweep(sim, n, model, dofs=range(18))
    root = sim.root_states.reshape(n, -1, 13)[:, 0].copy()
    expect = np.zeros(13, np.float32); expect[2] = 1.0; expect[6] = 1.0     # where dof_sweep placed it
    check_fixed_base_sweep(rows, np.tile(expect, (n, 1)), root)
