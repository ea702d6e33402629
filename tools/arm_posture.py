#!/usr/bin/env python3
"""Where do the checkpoint's joint means sit?  (the arm slots are 3 sigma off in every trained variant.)

The shipped checkpoint's observation normaliser (tests/golden/bez_kick_33_policy.npz: mean / variance of the 54 slots over the whole
PhysX training) against the ready pose, the joint limits, and the same statistics of THIS build's trained agents
(profiles/r04_fingerprint.json, made by tools/obs_fingerprint.py).  Also: how far the checkpoint's mean pose is from the
calf <-> foot-plate contact (tools/pair_penetration.py) -- the one unmodelled self-collision pair its ankle statistics sit against.

    python tools/arm_posture.py > profiles/r05_arm_posture.txt
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def ankle_gap(model, side, q4, q5):
    """Smallest signed distance (m) of the calf box's bottom corners to the top face of the foot plate, as a function of the ankle
    pitch q4 and the foot roll q5 (the joints are the links' own: axes and origins from the baked model)."""
    links = {l["name"]: l for l in model["links"]}
    boxes = {model["links"][b["link"]]["name"]: b for b in model["boxes"]}
    calf, ankle, foot = ("/%s_%s" % (side, p) for p in ("calve", "ankle", "foot"))

    def rot(axis, ang):
        a = np.asarray(axis, float); c, s = np.cos(ang), np.sin(ang)
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        return c * np.eye(3) + (1 - c) * np.outer(a, a) + s * K
    Ra, Rf = rot(links[ankle]["axis"], q4), rot(links[foot]["axis"], q5)
    R = Ra @ Rf                                                  # foot -> calf rotation
    o = np.asarray(links[ankle]["xyz"]) + Ra @ np.asarray(links[foot]["xyz"])   # foot origin in the calf frame
    cb, fb = boxes[calf], boxes[foot]
    ztop = fb["center"][2] + fb["half"][2]
    gaps = []
    for sx in (-1, 1):
        for sy in (-1, 1):
            p = np.array([cb["center"][0] + sx * cb["half"][0], cb["center"][1] + sy * cb["half"][1], cb["center"][2] - cb["half"][2]])
            gaps.append(float((R.T @ (p - o))[2] - ztop))
    return min(gaps)


def main():
    model = json.load(open(os.path.join(ROOT, "bez_isaacgym_amd", "model", "bez_model.json")))
    d = np.load(os.path.join(ROOT, "tests", "golden", "bez_kick_33_policy.npz"))
    m, v = d["running_mean_std/running_mean"].astype(np.float64), d["running_mean_std/running_var"].astype(np.float64)
    fp = json.load(open(os.path.join(ROOT, "profiles", "r04_fingerprint.json")))
    runs = [k for k in fp if k != "reference"]
    names, lo, hi, dflt = model["dof_names"], model["dof_lower"], model["dof_upper"], model["dof_default"]
    print("joint positions: checkpoint (PhysX, 4.03e9 samples) against the ready pose / limits, and this build's trained agents (z = (ours - ckpt) / ckpt sd)")
    print("%-18s %7s %6s | %6s %6s %6s | %s" % ("joint", "mean", "sd", "ready", "lower", "upper", "  ".join("%-18s" % r for r in runs)))
    for i in range(18):
        cells = []
        for r in runs:
            om, ov = fp[r]["mean"][i], fp[r]["var"][i]
            cells.append("%6.2f+-%4.2f z%+5.1f" % (om, np.sqrt(ov), (om - m[i]) / np.sqrt(v[i] + 1e-5)))
        print("%-18s %7.3f %6.3f | %6.2f %6.2f %6.2f | %s" % (names[i], m[i], np.sqrt(v[i]), dflt[i], lo[i], hi[i], "  ".join(cells)))
    print()
    print("arms: the checkpoint holds the elbows 0.29-0.37 rad beyond the ready pose's 1.5 rad (1.87 / 1.79, sd 0.21) and the shoulders 0.16 / 0.32 rad forward,")
    print("      1.3-1.7 rad from the nearest limit; no arm slot sits at a limit or at a shape: with the forearm's mesh bounds (y 0.0775..0.1265 m) clear of the")
    print("      torso's (|y| <= 0.0725 m) the arm never touches the torso, and forearm <-> hip_front never comes closer than 20 mm along the reference")
    print("      policy's rollouts (profiles/r05_pair_penetration.txt).  This build's agents park the arms elsewhere (z -4.3 / -3.8 on one slot per side):")
    print("      a different balance strategy of a different policy, not a posture a missing contact enforces.")
    print()
    print("ankles: smallest gap between the calf box's bottom corners and the foot plate's top face (same-leg grandparent pair, collision filter 0)")
    for side, i4, i5 in (("left", 8, 9), ("right", 16, 17)):
        g0 = ankle_gap(model, side, dflt[i4], dflt[i5])
        gm = ankle_gap(model, side, m[i4], m[i5])
        s4, s5 = np.sqrt(v[i4]), np.sqrt(v[i5])
        sgn = 1.0 if side == "left" else -1.0
        g1 = min(ankle_gap(model, side, m[i4] + s4, m[i5] + s5), ankle_gap(model, side, m[i4] + s4, m[i5] - s5))
        # roll at which the gap closes, at the checkpoint's mean pitch
        rr = np.linspace(0, 0.785, 400)
        closed = [r for r in rr if ankle_gap(model, side, m[i4], r) <= 0 or ankle_gap(model, side, m[i4], -r) <= 0]
        print("  %-5s ready pose %.1f mm | checkpoint mean pose (pitch %.2f, roll %+.2f) %.1f mm | mean + 1 sd on both %.1f mm | at the mean pitch the plate meets the calf at |roll| = %.2f rad (joint limit 0.785)"
              % (side, 1e3 * g0, m[i4], m[i5], 1e3 * gm, 1e3 * g1, closed[0] if closed else float("nan")))
        for r in runs:
            om4, om5 = fp[r]["mean"][i4], fp[r]["mean"][i5]
            print("        %-20s pitch %.2f roll %+.2f -> %.1f mm" % (r, om4, om5, 1e3 * ankle_gap(model, side, om4, om5)))


if __name__ == "__main__":
    main()
