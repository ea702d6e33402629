#!/usr/bin/env python3
"""Build-container only: golden vectors for the cleats feet sensor and the bez_walk / bez_orient task logic, produced by
the REFERENCE's own TorchScript functions imported under the same stub as tests/golden/make_golden.py:

  * tasks.kick_env.compute_feet_sensors_cleats                      kick_env.py:1044-1069
  * tasks.walk_env.compute_bez_reward / compute_bez_observations     walk_env.py:826-1050
  * tasks.orient_env.compute_off_angle / compute_bez_reward          orient_env.py:719-735, 843-1018

Output: tests/golden/tasks_golden.npz (inputs + outputs, numbers only)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

OUT = os.path.join(HERE, "tasks_golden.npz")


def main():
    import json
    MG.install_stub()
    import torch
    import tasks.kick_env as K
    import tasks.walk_env as W
    import tasks.orient_env as O
    model = json.load(open(os.path.join(HERE, "..", "..", "bez_isaacgym_amd", "model", "bez_model.json")))
    rng = np.random.default_rng(20261004)
    T = torch.from_numpy
    G, N = {}, 64
    default = np.tile(np.array(model["dof_default"], np.float32), (N, 1))
    up_vec = np.tile(np.array([[0, 0, 1.0]], np.float32), (N, 1))

    # ---- cleats feet sensor: forces around the 1 N threshold on the 4 + 4 cleat rows
    lf = (rng.normal(size=(N, 4, 3)) * 1.2).astype(np.float32)
    rf = (rng.normal(size=(N, 4, 3)) * 1.2).astype(np.float32)
    lf[:8] = 0.0; rf[8:16] *= 10.0
    lf[16, 0] = [0.6, 0.0, 0.7998]     # |f| = 0.99984: just below the 1 N gate
    lf[17, 1] = [0.0, 0.0, 1.0002]     # just above
    out = K.compute_feet_sensors_cleats(T(lf), T(rf), torch.tensor([[-1.] * 8]).repeat((N, 1)), torch.ones((N, 8)))
    G.update(cleats_left=lf, cleats_right=rf, cleats_out=out.numpy())

    def scene(tag, mod, task):
        q = MG.rand_quat(rng, N, tilt=0.25)
        root = (rng.normal(size=(N, 3)) * 0.3).astype(np.float32); root[:, 2] = 0.32
        v = (rng.normal(size=(N, 3)) * 0.3).astype(np.float32); w = (rng.normal(size=(N, 3)) * 1.0).astype(np.float32)
        dof = default + (rng.normal(size=(N, 18)) * 0.2).astype(np.float32)
        dofv = (rng.normal(size=(N, 18)) * 2.0).astype(np.float32)
        goal = rng.uniform(-2, 2, size=(N, 2)).astype(np.float32)
        reset = (rng.random(N) < 0.1).astype(np.int64)
        progress = rng.integers(1, 590, size=N).astype(np.int64)
        feet = rng.choice(np.array([-1.0, 1.0], np.float32), size=(N, 8)).astype(np.float32)
        mod(dict(q=q, root=root, v=v, w=w, dof=dof, dofv=dofv, goal=goal, reset=reset, progress=progress))
        bez_init = np.zeros(2, np.float32)
        if task == "walk":
            rew, rst = W.compute_bez_reward(T(dof), T(default), T(v), T(w), T(root), T(q), T(up_vec), T(goal), T(reset), T(progress),
                                            T(feet), T(bez_init.copy()), 600, N, 0.01667, False)
            off = W.compute_off_orn(T(root), T(q), T(goal))
        else:
            ga = np.full((N, 1), 1.5708, np.float32)
            rew, rst = O.compute_bez_reward(T(dof), T(default), T(v), T(w), T(root), T(q), T(up_vec), T(ga), T(reset), T(progress),
                                            T(feet), T(bez_init.copy()), 600, N, 0.01667, False)
            off = O.compute_off_angle(T(q), T(ga))
        imu = (rng.normal(size=(N, 6))).astype(np.float32)
        obs = W.compute_bez_observations(T(dof), T(dofv), T(imu), off, T(feet))
        assert obs.shape == (N, 52)
        for k, val in dict(q=q, root=root, v=v, w=w, dof=dof, dofv=dofv, goal=goal, reset=reset, progress=progress, feet=feet,
                           rew=rew.numpy(), rst=rst.numpy(), off=off.numpy()).items():
            G["%s_%s_%s" % (task, tag, k)] = val

    def normal(d):
        pass

    def edge_walk(d):
        d["q"][0:6] = [[0.0, 0.6, 0.0, 0.8]] * 6                 # fallen: up_proj < 0.7
        d["root"][6:10, 0:2] = d["goal"][6:10] + 0.01             # at the goal ...
        d["dof"][6:8] = default[6:8] + 0.01; d["v"][6:8] = 0.01; d["w"][6:8] = 0.01   # ... and still: win state
        d["root"][10:14, 0:2] = -0.5 * d["goal"][10:14]           # walked away: heading to goal flipped by > 90 deg
        d["progress"][14:18] = [599, 600, 601, 650]               # horizon
        d["progress"][6:8] = [10, 600]                            # win + horizon ordering
        d["reset"][18:22] = 1

    def edge_orient(d):
        d["q"][0:6] = [[0.0, 0.6, 0.0, 0.8]] * 6
        for i, yaw in enumerate([1.5708, 1.53, 1.60, 1.50, -1.5, 3.1, -3.1, 0.0]):  # heading error around the 0.05 threshold, both signs
            d["q"][6 + i] = [0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2)]
        d["dof"][6:9] = default[6:9] + 0.01; d["v"][6:9] = 0.01; d["w"][6:9] = 0.01   # win state where also aligned
        d["root"][14:18, 0:2] = [[0.31, 0.0], [0.0, -0.31], [0.2, 0.2], [0.29, 0.0]]     # wandered > 0.3 m
        d["progress"][18:22] = [599, 600, 601, 650]
        d["reset"][22:26] = 1

    scene("normal", normal, "walk")
    scene("edge", edge_walk, "walk")
    scene("normal", normal, "orient")
    scene("edge", edge_orient, "orient")
    np.savez_compressed(OUT, **G)
    print("wrote", OUT, len(G), "arrays")


if __name__ == "__main__":
    main()
