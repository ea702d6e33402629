#!/usr/bin/env python3
"""the reference's shipped policy in the CPU oracle under the TGS-shaped unified substep
(BEZ_FLAG_TGS_SOLVER, oracle/bez_oracle_tgs.inc) and every knob of it, one table.  Judged by (i) the policy's sim-to-sim
(goal rate / return / episode length; PhysX: 87.55, goals after ~110 steps) and (ii) the rollout's observation statistics against
the checkpoint's running mean / variance: per-joint speed sigma (arms 3.8-4.2 rad/s, legs 2.0-3.7 under PhysX) and feet-flag means.

    python tools/s2s_tgs_sweep.py > profiles/r04_s2s_tgs.md        (CPU only; ~10 s per row)
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import s2s_cpu as S  # noqa: E402

T = 513  # BEZ_FLAG_IMU_PREV_ALIAS | BEZ_FLAG_TGS_SOLVER
VARIANTS = [
    ("compliant model (the shipped kernels' physics)", {}, None, {}),
    ("rigid contact of round 3 (PGS after the implicit-drive ABA)", {}, 65, {}),
    ("TGS, every knob at its default", {}, T, {}),
    ("TGS, leg<->leg as the compliant spring (18=1)", {}, T, {18: 1}),
    ("TGS, effort limit read as an impulse per substep (10=1)", {}, T, {10: 1, 18: 1}),
    ("TGS, effort x2 (10=2)", {}, T, {10: 2, 18: 1}),
    ("TGS, joint friction = 0.1 x unconstrained-stage joint force (11=1)", {}, T, {11: 1, 18: 1}),
    ("TGS, no joint friction (11=3)", {}, T, {11: 3, 18: 1}),
    ("TGS, joint friction = 0.1 x force the joint really transmitted (11=4)", {}, T, {11: 4, 18: 1}),
    ("TGS, same, force part only (11=5)", {}, T, {11: 5, 18: 1}),
    ("TGS, same, clamp per sub-step (11=6)", {}, T, {11: 6, 18: 1}),
    ("TGS, speed limit also a solver row (12=1)", {}, T, {12: 1, 18: 1}),
    ("TGS, speed limit + hard clamp of the final rates (12=2)", {}, T, {12: 2, 18: 1}),
    ("TGS, no speed limit at all (12=3)", {}, T, {12: 3, 18: 1}),
    ("TGS, 1 position iteration", {}, T, {8: 1, 18: 1}),
    ("TGS, 2 position iterations", {}, T, {8: 2, 18: 1}),
    ("TGS, 8 position iterations", {}, T, {8: 8, 18: 1}),
    ("TGS, 16 position iterations", {}, T, {8: 16, 18: 1}),
    ("TGS, no velocity iteration", {}, T, {9: -1, 18: 1}),
    ("TGS, 4 velocity iterations", {}, T, {9: 4, 18: 1}),
    ("TGS, 4 substeps", {"substeps": 4}, T, {18: 1}),
    ("TGS, penetration ERP 0.2, depenetration <= 1 m/s", {}, T, {13: 0.2, 14: 1.0, 18: 1}),
    ("TGS, contact margin 5 mm", {}, T, {15: 0.005, 18: 1}),
    ("TGS, velocity-level friction only (16=1)", {}, T, {16: 1, 18: 1}),
    ("TGS, joints before contacts in a sweep (17=1)", {}, T, {17: 1, 18: 1}),
    ("TGS, drives damper-only in the velocity iteration (19=1)", {}, T, {19: 1, 18: 1}),
    ("TGS, drive error frozen over the substep (20=1)", {}, T, {20: 1, 18: 1}),
    ("TGS, real-load joint friction + speed row (11=4 12=1)", {}, T, {11: 4, 12: 1, 18: 1}),
    ("TGS, real-load joint friction + 8 iterations", {}, T, {11: 4, 8: 8, 18: 1}),
    ("TGS, real-load joint friction, coefficient 0.05", {"joint_friction": 0.05}, T, {11: 4, 18: 1}),
    ("TGS, real-load joint friction, coefficient 0.2", {"joint_friction": 0.2}, T, {11: 4, 18: 1}),
    ("TGS, damping 2 (the round-3 training lever)", {"kd": 2.0}, T, {18: 1}),
    ("TGS, effort 1.5 N m", {"effort": 1.5}, T, {18: 1}),
    ("TGS, effort 5 N m", {"effort": 5.0}, T, {18: 1}),
]


def main():
    pol = S.NumpyPolicy()
    ref_sd = np.sqrt(pol.var)
    rows = []
    print("checkpoint (PhysX): joint-speed sigma arms %s, left leg %s, right leg %s; feet-flag means %s\n" % (
        np.round(ref_sd[[20, 21, 28, 29]], 2), np.round(ref_sd[22:28], 2), np.round(ref_sd[30:36], 2), np.round(pol.mean[44:52], 2)))
    print("| variant | goal rate | mean return | mean length | goal length | fall rate | obs z-rms | z[43] | speed sigma / checkpoint's: arms, legs | feet flag 0 / 4 mean |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for name, over, flags, tune in VARIANTS:
        r = S.evaluate(pol, over, flags, tune, n=256, steps=600, seed=1, stochastic=True)
        e = max(r["episodes"], 1)
        sr = np.array(r["obs_std_ratio"])
        z = np.array(r["obs_z"])
        feet = pol.mean[[44, 48]] + z[[44, 48]] * np.sqrt(pol.var[[44, 48]] + 1e-5)
        rows.append(dict(variant=name, overrides=over, flags=flags, tune={str(k): v for k, v in tune.items()},
                         **{k: r[k] for k in ("episodes", "goal_rate", "mean_return", "mean_length", "goal_length", "reasons", "obs_z", "obs_std_ratio", "obs_z_rms")}))
        print("| %s | %.3f | %.2f | %.1f | %.0f | %.3f | %.2f | %.1f | %.2f, %.2f | %.2f / %.2f |" % (
            name, r["goal_rate"], r["mean_return"], r["mean_length"], r["goal_length"], r["reasons"]["fall"] / e, r["obs_z_rms"], z[43],
            sr[[20, 21, 28, 29]].mean(), sr[[22, 23, 24, 25, 26, 27, 30, 31, 32, 33, 34, 35]].mean(), feet[0], feet[1]), flush=True)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r04_s2s_tgs.json"), "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
