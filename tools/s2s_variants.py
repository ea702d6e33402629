#!/usr/bin/env python3
"""the reference's shipped policy in the CPU oracle under every contact / actuation variant that was
tried, one table (goal rate, mean return, episode length, fall rate, z-distance of obs[36:44] to the checkpoint's statistics).
PhysX: return 87.55, ~110 steps, goals essentially always.   python tools/s2s_variants.py > profiles/r03_s2s_variants.md"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import s2s_cpu as S  # noqa: E402

H = 65  # BEZ_FLAG_IMU_PREV_ALIAS | BEZ_FLAG_HARD_CONTACT
VARIANTS = [
    ("compliant contact, round-2 defaults", {}, None, {}),
    ("compliant, ball-only damping ratio 1 (ball_cn 155)", {"ball_cn": 155.0}, None, {}),
    ("compliant, ball_cn 400", {"ball_cn": 400.0}, None, {}),
    ("compliant, all contacts critically damped (contact_cn 240)", {"contact_cn": 240.0}, None, {}),
    ("compliant, stick viscosity 1e5 N s/m, v_eps 0.5 mm/s (no creep)", {"contact_cn": 1000.0, "contact_ct": 1e5, "contact_veps": 5e-4}, None, {}),
    ("RIGID contact (PGS, stiction, restitution 0)", {}, H, {}),
    ("rigid, 8 substeps", {"substeps": 8}, H, {}),
    ("rigid + joint speed limit as a constraint", {}, H, {7: 1, 5: 6}),
    ("rigid + speed limit as a constraint, 8 substeps", {"substeps": 8}, H, {7: 1, 5: 6}),
    ("rigid + speed limit on incoming rates only", {}, H, {7: 2, 5: 6}),
    ("rigid + shape rest offsets (robot 1 cm = asset thickness, ball 2 cm)", {}, H, {7: 1, 5: 6, 6: 0.01}),
    ("rigid, effort 1.5 N m", {"effort": 1.5}, H, {}),
    ("rigid, effort 5 N m", {"effort": 5.0}, H, {}),
    ("rigid, effort 300 N m (limit read as an impulse)", {"effort": 300.0}, H, {}),
    ("rigid, effort 300 + speed limit 100 rad/s", {"effort": 300.0, "vel_limit": 100.0}, H, {}),
    ("rigid, joint friction 1 N m with stiction (v_eps 0.02)", {"joint_friction": 1.0, "jfric_veps": 0.02}, H, {}),
    ("rigid, best of 400 random draws by goal rate", {"effort": 2.0027, "vel_limit": 14.5065, "kp": 55.9443, "kd": 3.7685, "armature": 0.0022, "jfric_veps": 0.2415,
                                                      "limit_k": 2539.3, "limit_d": 26.389, "self_kn": 11239.8, "self_cn": 3.8964, "joint_friction": 0.2908,
                                                      "plane_friction": 1.3131, "ball_ang_damping": 0.1388, "substeps": 4}, H, {1: 0.1783, 2: 0.0041, 3: 0.1017, 0: 6, 7: 2}),
    ("rigid, best of 400 random draws by episode length", {"effort": 3.9483, "vel_limit": 4.078, "kp": 141.54, "kd": 5.8155, "armature": 0.009, "jfric_veps": 0.0449,
                                                           "limit_k": 324.58, "limit_d": 10.706, "self_kn": 8995.0, "self_cn": 14.746, "joint_friction": 0.4157,
                                                           "plane_friction": 1.4369, "ball_ang_damping": 0.7226}, H, {1: 0.3963, 2: 0.0192, 3: 0.1018, 0: 17, 6: 0.0043}),
]


def main():
    pol = S.NumpyPolicy()
    rows = []
    print("| variant | goal rate | mean return | mean length | fall rate | obs_z[36:44] (imu 0-5, |sin|, -cos) |")
    print("|---|---|---|---|---|---|")
    for name, over, flags, tune in VARIANTS:
        r = S.evaluate(pol, over, flags, tune, n=256, steps=600, seed=1)
        e = max(r["episodes"], 1)
        rows.append(dict(variant=name, overrides=over, flags=flags, tune={str(k): v for k, v in tune.items()}, **{k: r[k] for k in ("episodes", "goal_rate", "mean_return", "mean_length", "reasons", "obs_z")}))
        print("| %s | %.3f | %.2f | %.1f | %.3f | %s |" % (name, r["goal_rate"], r["mean_return"], r["mean_length"], r["reasons"]["fall"] / e,
                                                          " ".join("%.1f" % v for v in r["obs_z"][36:44])), flush=True)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r03_s2s_variants.json"), "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
