#!/usr/bin/env python3
"""Random search over the free constants AND the structural switches of the oracle's rigid-contact model, scored by the
reference policy's sim-to-sim (tools/s2s_cpu.py).  Build-container experiment; appends one JSON line per draw.

    OMP_NUM_THREADS=4 python tools/s2s_search.py --draws 300 --out gpurun_out/r03_s2s_search.jsonl
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import s2s_cpu as S  # noqa: E402

LOG = {"effort": (1.5, 5.0), "vel_limit": (4.0, 15.0), "kp": (50.0, 200.0), "kd": (3.0, 15.0), "armature": (3e-4, 1e-2),
       "jfric_veps": (0.02, 0.5), "limit_k": (50.0, 5e3), "limit_d": (0.5, 50.0), "self_kn": (5e2, 3e4), "self_cn": (1.0, 50.0)}
LIN = {"joint_friction": (0.0, 0.5), "plane_friction": (0.5, 1.5), "ball_ang_damping": (0.0, 1.0)}
TUNE_LOG = {1: (0.05, 0.8), 2: (0.002, 0.02), 3: (0.1, 10.0)}


def draw(rng):
    o = {k: math.exp(rng.uniform(math.log(a), math.log(b))) for k, (a, b) in LOG.items()}
    o.update({k: rng.uniform(a, b) for k, (a, b) in LIN.items()})
    o["substeps"] = int(rng.choice([2, 2, 4]))
    t = {k: math.exp(rng.uniform(math.log(a), math.log(b))) for k, (a, b) in TUNE_LOG.items()}
    t[0] = float(rng.integers(4, 31))
    t[7] = float(rng.integers(0, 3))
    t[6] = float(rng.choice([0.0, 0.0, rng.uniform(0.0, 0.01)]))
    return o, t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--draws", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--envs", type=int, default=128)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--out", default="gpurun_out/r03_s2s_search.jsonl")
    a = ap.parse_args()
    pol = S.NumpyPolicy()
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    with open(a.out, "a") as f:
        for i in range(a.draws):
            o, t = ({}, {}) if i == 0 else draw(rng)
            r = S.evaluate(pol, o, 65, t, a.envs, a.steps, seed=1)
            rec = dict(i=i, over=o, tune=t, goal_rate=r["goal_rate"], mean_return=r["mean_return"], mean_length=r["mean_length"],
                       reasons=r["reasons"], z43=r["obs_z"][43], zrms=r["obs_z_rms"])
            f.write(json.dumps(rec) + "\n"); f.flush()
            if i % 10 == 0:
                print("[%d/%d] %.0f s  len %.1f goal %.3f" % (i, a.draws, time.time() - t0, r["mean_length"], r["goal_rate"]), flush=True)


if __name__ == "__main__":
    main()
