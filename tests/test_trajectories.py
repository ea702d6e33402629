"""The scripted-input harness: key-frame playback of the reference's `simulation_*.csv` trajectories
(resources/library/trajectories/src/soccer_trajectories.py, scenario bez_isaacgym/test/test_kick_env.py:210-222) as a
known-answer test of the physics: the open-loop right kick must send the ball forward while the robot stays on its feet."""
import json
import os

import numpy as np
import pytest

from bez_isaacgym_amd import abi
from bez_isaacgym_amd.utils.trajectories import JOINT_ORDER, Trajectory, read_csv_table

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLES = os.path.join(ROOT, "tests", "golden", "trajectories.json")
REF_CSV = "/root/reference/resources/library/trajectories/trajectories/simulation_rightkick.csv"


def _tables():
    return json.load(open(TABLES))


def test_interpolation_and_clock(model):
    """Ready pose prepended at t=0 and appended 1 s after the last key; linear interpolation; 0.00833 s of trajectory per
    env step; action = position - default pose (soccer_trajectories.py:36-44,61-90)."""
    ready = dict(zip(model["dof_names"], model["dof_default"]))
    tr = Trajectory(_tables()["simulation_rightkick"], ready)
    assert tr.max_time == pytest.approx(3.0) and list(model["dof_names"]) == JOINT_ORDER
    sp0, sp_end = tr.get_setpoint(0.0), tr.get_setpoint(3.0)
    assert sp0["right_leg_motor_3"] == pytest.approx(-1.176) and sp_end["left_arm_motor_1"] == pytest.approx(1.5)
    # half-way between t=0 (ready -1.176) and the first key at 0.69 s (-0.4)
    assert tr.get_setpoint(0.345)["right_leg_motor_3"] == pytest.approx((-1.176 - 0.4) / 2, abs=1e-9)
    assert tr.get_setpoint(1.41)["right_leg_motor_2"] == pytest.approx(1.5)     # the kick key-frame
    with pytest.raises(ValueError):
        tr.get_setpoint(3.01)
    acts = tr.actions(model["dof_default"])
    assert acts.shape == (361, 18) and np.abs(acts[0]).max() < 1e-6
    k = int(np.ceil(1.41 / 0.00833))  # first clock tick past the kick key-frame
    assert acts[k, JOINT_ORDER.index("right_leg_motor_2")] == pytest.approx(1.5 - 0.564, abs=3e-2)


@pytest.mark.skipif(not os.path.exists(REF_CSV), reason="reference tree only exists in the build container")
def test_fixture_equals_the_reference_csv():
    assert read_csv_table(REF_CSV) == _tables()["simulation_rightkick"]


def _play(stepper, n, model):
    ready = dict(zip(model["dof_names"], model["dof_default"]))
    acts = Trajectory(_tables()["simulation_rightkick"], ready).actions(model["dof_default"])
    best_x = np.full(n, 0.175)
    min_z = np.full(n, 1.0)
    alive = np.ones(n, bool)
    for k, a in enumerate(acts[:300]):
        rs, reset = stepper(np.tile(a, (n, 1)))
        best_x = np.where(alive, np.maximum(best_x, rs[:, 1, 0]), best_x)
        min_z = np.where(alive, np.minimum(min_z, rs[:, 0, 2]), min_z)
        fell = (reset > 0) & (rs[:, 0, 2] < 0.275)
        assert not fell.any(), "robot fell during the scripted kick at step %d" % k
        alive &= ~(reset > 0)   # an env that finished its episode (ball past the goal line) is reset: stop tracking it
    return best_x, min_z


def test_rightkick_moves_the_ball_forward_oracle(model):
    from oracle.bez_oracle import Oracle
    n = 128
    o = Oracle(abi.default_config(n, seed=5))

    def stepper(a):
        o.step(a)
        return o.root_states.reshape(n, 2, 13), o.reset_buf
    best_x, min_z = _play(stepper, n, model)
    # every env starts from its own reset draw (joint noise +-0.15 rad): the kick connects well in most of them
    assert (best_x - 0.175 > 0.3).mean() >= 0.9 and np.median(best_x - 0.175) > 0.8, np.sort(best_x)
    assert (min_z > 0.30).all(), min_z            # and the robot stayed on its feet


@pytest.mark.gpu
def test_rightkick_moves_the_ball_forward_hip(model):
    """Same scenario through KickEnv.step (the reference drives env.step too), 256 HIP envs."""
    import torch
    from bez_isaacgym_amd.tasks import isaacgym_task_map
    from bez_isaacgym_amd.utils.config import load_config
    from bez_isaacgym_amd.utils.trajectories import SoccerTrajectoryClass
    n = 256
    cfg = load_config(["task=bez_kick", "num_envs=%d" % n, "headless=True"])
    task_cfg = cfg["task"]
    task_cfg["rl_device"] = "cuda:0"
    env = isaacgym_task_map["bez_kick"](cfg=task_cfg, sim_device="cuda:0", graphics_device_id=0, headless=True)
    env.reset()
    best_x = torch.full((n,), 0.175, device="cuda:0")
    alive = torch.ones(n, dtype=torch.bool, device="cuda:0")
    state = {"fell": 0}

    def on_step(k, res):
        nonlocal best_x, alive
        if k >= 300:
            return
        rs = env.root_states.view(n, 2, 13)
        done = env.reset_buf > 0
        state["fell"] += int((done & (rs[:, 0, 2] < 0.275) & alive).sum())
        best_x = torch.where(alive, torch.maximum(best_x, rs[:, 1, 0]), best_x)
        alive &= ~done
    SoccerTrajectoryClass(env, 0, TABLES).run_trajectory("rightkick", on_step)
    assert state["fell"] <= n // 50, state   # (round 6: 2 of 256 with the stiff leg<->leg contact; 0 in the oracle's 16)
    d = (best_x - 0.175).cpu().numpy()
    assert (d > 0.3).mean() >= 0.9 and np.median(d) > 0.8, np.sort(d)[:10]
