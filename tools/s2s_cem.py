#!/usr/bin/env python3
"""Cross-entropy search (diagonal Gaussian in log space, elite fraction 1/4) over the rigid-contact model's constants, scored by
the reference policy's sim-to-sim in the oracle: score = 100 * goal rate + mean episode length / 10.  Continues where the random
search (tools/s2s_search.py) stopped.   OMP_NUM_THREADS=4 python tools/s2s_cem.py --gens 12 --pop 24"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import s2s_cpu as S  # noqa: E402

# name -> (low, high, start); all searched in log space except those marked linear
P = {"effort": (1.2, 6.0, 2.5), "vel_limit": (3.0, 20.0, 6.283), "kp": (30.0, 300.0, 100.0), "kd": (2.0, 20.0, 7.5), "armature": (2e-4, 2e-2, 1e-3),
     "joint_friction": (0.01, 1.0, 0.1), "jfric_veps": (0.01, 1.0, 0.1), "plane_friction": (0.4, 2.0, 1.0), "limit_k": (30.0, 1e4, 200.0),
     "limit_d": (0.3, 100.0, 2.0), "ball_ang_damping": (0.02, 2.0, 0.5), "self_kn": (3e2, 5e4, 3e3), "self_cn": (0.5, 100.0, 5.0),
     "t1": (0.03, 0.9, 0.2), "t2": (0.001, 0.03, 0.02), "t3": (0.05, 20.0, 1.0)}
NAMES = list(P)


def decode(x, mode):
    over, tune = {}, {0: 16.0, 5: 4.0, 7: float(mode)}
    for n, v in zip(NAMES, x):
        lo, hi, _ = P[n]
        val = min(max(math.exp(v), lo), hi)
        if n.startswith("t"):
            tune[int(n[1:])] = val
        else:
            over[n] = val
    return over, tune


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gens", type=int, default=12)
    ap.add_argument("--pop", type=int, default=24)
    ap.add_argument("--envs", type=int, default=128)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--mode", type=int, default=1, help="tune[7]: joint-speed-limit treatment")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default="gpurun_out/r03_s2s_cem.jsonl")
    a = ap.parse_args()
    pol = S.NumpyPolicy()
    rng = np.random.default_rng(a.seed)
    mean = np.array([math.log(P[n][2]) for n in NAMES])
    std = np.array([(math.log(P[n][1]) - math.log(P[n][0])) / 6.0 for n in NAMES])
    t0 = time.time()
    with open(a.out, "a") as f:
        for g in range(a.gens):
            xs = [mean] + [mean + std * rng.standard_normal(len(NAMES)) for _ in range(a.pop - 1)]
            scored = []
            for x in xs:
                over, tune = decode(x, a.mode)
                r = S.evaluate(pol, over, 65, tune, a.envs, a.steps, seed=1 + g)
                sc = 100.0 * r["goal_rate"] + r["mean_length"] / 10.0
                scored.append((sc, x))
                f.write(json.dumps(dict(gen=g, score=sc, goal_rate=r["goal_rate"], mean_length=r["mean_length"], mean_return=r["mean_return"], over=over,
                                        tune={str(k): v for k, v in tune.items()})) + "\n"); f.flush()
            scored.sort(key=lambda t: -t[0])
            elite = np.array([x for _, x in scored[:max(a.pop // 4, 2)]])
            mean = 0.5 * mean + 0.5 * elite.mean(0)
            std = np.maximum(0.6 * std + 0.4 * elite.std(0), 0.03)
            print("[gen %d] %.0f s  best %.2f  mean-of-elite %.2f  centre %.2f" % (g, time.time() - t0, scored[0][0], np.mean([s for s, _ in scored[:len(elite)]]),
                                                                                  [s for s, x in scored if x is xs[0]][0]), flush=True)


if __name__ == "__main__":
    main()
