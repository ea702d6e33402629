"""Open-loop key-frame trajectories (get-up, kicks) played through env.step -- the reference's scripted-input harness
(resources/library/trajectories/src/soccer_trajectories.py:12-118, scenario bez_isaacgym/test/test_kick_env.py:210-222).

Restated behaviour: a trajectory is a table joint -> values at key times; the ready pose (cfg env.readyJointAngles) is
prepended at t = 0 and appended one second after the last key (soccer_trajectories.py:36-44); joints are interpolated
linearly (scipy interp1d default); `publish` advances the trajectory clock by 0.00833 s per env.step() and sends
`position - default_dof_pos` as the action (soccer_trajectories.py:61-88) -- with the 1/60 s control step of
bez_kick(_test).yaml the motion therefore plays at half speed in simulated time, as it does in the reference.
Tables come from the reference's CSV format (first column joint name, `time` and `comment` rows) or from the numeric
fixture tests/golden/trajectories.json."""
import csv
import json

import numpy as np
import torch

JOINT_ORDER = ["head_motor_0", "head_motor_1", "left_arm_motor_0", "left_arm_motor_1",
               "left_leg_motor_0", "left_leg_motor_1", "left_leg_motor_2", "left_leg_motor_3", "left_leg_motor_4", "left_leg_motor_5",
               "right_arm_motor_0", "right_arm_motor_1",
               "right_leg_motor_0", "right_leg_motor_1", "right_leg_motor_2", "right_leg_motor_3", "right_leg_motor_4", "right_leg_motor_5"]
TRAJECTORY_CLOCK_STEP = 0.00833  # soccer_trajectories.py:90


def read_csv_table(path):
    """The reference's CSV layout -> {"time": [...], "joints": {name: [...]}} (comment rows dropped)."""
    times, joints = None, {}
    with open(path) as f:
        for row in csv.reader(f):
            if not row or row[0] == "comment":
                continue
            if row[0] == "time":
                times = [float(x) for x in row[1:]]
            else:
                joints[row[0]] = [float(x) for x in row[1:]]
    return {"time": times, "joints": joints}


class Trajectory:
    """Interpolates a key-frame table for multiple joints (soccer_trajectories.py:12-55)."""

    def __init__(self, table, ready_joint_angles, mirror=False, time_to_last_pose=1.0):
        self.mirror = mirror
        t = list(table["time"])
        self.times = np.array([0.0] + t + [t[-1] + time_to_last_pose])
        self.max_time = float(self.times[-1])
        self.values = {}
        for name, vals in table["joints"].items():
            ready = float(ready_joint_angles[name])
            self.values[name] = np.array([ready] + list(vals) + [ready])

    def joints(self):
        return self.values.keys()

    def get_setpoint(self, timestamp):
        if timestamp < self.times[0] or timestamp > self.times[-1]:
            raise ValueError("timestamp outside the trajectory")
        return {j: float(np.interp(timestamp, self.times, v)) for j, v in self.values.items()}

    def position(self, timestamp):
        """18 joint positions in DOF order (joints the table does not name stay 0, soccer_trajectories.py:76)."""
        pos = [0.0] * 18
        for j, sp in self.get_setpoint(timestamp).items():
            pos[JOINT_ORDER.index(j)] = sp
        if self.mirror:  # soccer_trajectories.py:82-88, as written there
            m = list(pos)
            m[0:2] = pos[2:4]; m[2:4] = pos[0:2]; m[4:10] = pos[10:16]; m[10:16] = pos[4:10]
            pos = m
        return pos

    def actions(self, default_dof_pos):
        """All actions of one playback, (steps, 18): position(t) - default pose for t = 0, 0.00833, ... < max_time."""
        out, t = [], 0.0
        d = np.asarray(default_dof_pos, dtype=np.float64)
        while t < self.max_time:
            out.append(np.asarray(self.position(t)) - d)
            t += TRAJECTORY_CLOCK_STEP
        return np.asarray(out, dtype=np.float32)

    def publish(self, env, on_step=None):
        """Play the trajectory in every env of `env` (the reference drives its single test env)."""
        default = env.default_dof_pos[0].detach().cpu().numpy()
        for k, a in enumerate(self.actions(default)):
            action = torch.as_tensor(a, dtype=torch.float, device=env.device).unsqueeze(0).repeat(env.num_envs, 1)
            res = env.step(action)
            if on_step is not None:
                on_step(k, res)


class SoccerTrajectoryClass:
    """soccer_trajectories.py:94-113: run_trajectory("rightkick") plays `simulation_rightkick`."""

    def __init__(self, env, env_ids=None, tables=None):
        self.env, self.env_ids = env, env_ids
        if isinstance(tables, str):
            with open(tables) as f:
                tables = json.load(f)
        self.tables = tables or {}

    def run_trajectory(self, command, on_step=None):
        name = "simulation_" + command
        if name not in self.tables:
            return None
        ready = self.env.cfg["env"]["readyJointAngles"]
        traj = Trajectory(self.tables[name], ready, False)
        traj.publish(self.env, on_step)
        return traj
