#!/usr/bin/env python3
"""Which self-collision pairs does `collision_filter 0` enable that this build does not model?  

`kick_env.py:365-366` creates the robot with collision filter 0: PhysX collides every pair of its shapes except parent / child pairs
of the articulation.  The build models the ten leg boxes as capsules, left x right only (`bez_model.json: capsule_pairs`).  This tool
plays the reference's shipped policy (numeric fixture tests/golden/bez_kick_33_policy.npz) in the CPU oracle, takes the Isaac-layout
rigid-body rows of every control step and measures, for EVERY non-adjacent pair of the URDF's collision shapes
(oracle/collision_shapes.json: boxes as written, mesh links as the bounding box of their vertices), the separation by the separating-axis
test of two oriented boxes (15 axes; negative = penetration depth).  Reported per pair: share of the samples in which it penetrates, share
within 0.02 m (two shapes resting `urdfAsset.thickness` = 0.01 apart, bez_kick.yaml:88), the deepest penetration.

Build-container experiment harness (loads the oracle); never on the product path.

    python tools/pair_penetration.py --steps 60 --out profiles/r05_pair_penetration.json
"""
import argparse
import itertools
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bez_isaacgym_amd import abi  # noqa: E402


def quat_to_mat(q):
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z); R[..., 0, 1] = 2 * (x * y - z * w); R[..., 0, 2] = 2 * (x * z + y * w)
    R[..., 1, 0] = 2 * (x * y + z * w); R[..., 1, 1] = 1 - 2 * (x * x + z * z); R[..., 1, 2] = 2 * (y * z - x * w)
    R[..., 2, 0] = 2 * (x * z - y * w); R[..., 2, 1] = 2 * (y * z + x * w); R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def obb_separation(ca, Ra, ha, cb, Rb, hb):
    """Separating-axis test of two oriented boxes, batched over the leading axis.  Returns max over the 15 axes of
    |t.L| - (ra + rb): > 0 separated (a lower bound of the distance), < 0 overlapping by that depth."""
    t = cb - ca
    best = np.full(t.shape[0], -np.inf)
    axes = [Ra[:, :, i] for i in range(3)] + [Rb[:, :, i] for i in range(3)]
    for i in range(3):
        for j in range(3):
            axes.append(np.cross(Ra[:, :, i], Rb[:, :, j]))
    for L in axes:
        nrm = np.linalg.norm(L, axis=1)
        ok = nrm > 1e-6
        Ln = L / np.maximum(nrm, 1e-12)[:, None]
        ra = sum(ha[i] * np.abs(np.einsum("nk,nk->n", Ra[:, :, i], Ln)) for i in range(3))
        rb = sum(hb[i] * np.abs(np.einsum("nk,nk->n", Rb[:, :, i], Ln)) for i in range(3))
        s = np.abs(np.einsum("nk,nk->n", t, Ln)) - ra - rb
        best = np.where(ok, np.maximum(best, s), best)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=256)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--asset", default="stl", choices=("stl", "box"))
    ap.add_argument("--policy", default="reference", choices=("reference", "zero", "random"))
    ap.add_argument("--flags", type=int, default=None)
    ap.add_argument("--stochastic", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("--set", nargs="*", default=[], help="BezSimConfig overrides, key=value")
    ap.add_argument("--modelled-only", action="store_true")
    ap.add_argument("--skip", type=int, default=10, help="control steps after a reset that are left out (a third of the reset draws START with the leg capsules overlapping by up to 2 cm: +-0.15 rad on the hip rolls)")
    a = ap.parse_args()
    from oracle.bez_oracle import Oracle
    from s2s_cpu import NumpyPolicy
    model = json.load(open(os.path.join(ROOT, "bez_isaacgym_amd", "model", "bez_model.json")))
    shapes = json.load(open(os.path.join(ROOT, "oracle", "collision_shapes.json")))[a.asset]
    names = model["body_names"]
    body_parent = {}
    for l in model["links"]:
        if l["parent"] >= 0:
            body_parent[l["body"]] = model["links"][l["parent"]]["body"]
    bodies = [i for i, nm in enumerate(names) if nm in shapes]
    modelled = set()
    caps = model["capsules"]
    for pa, pb in model["capsule_pairs"]:
        ba, bb = model["links"][caps[pa]["link"]]["body"], model["links"][caps[pb]["link"]]["body"]
        modelled.add((min(ba, bb), max(ba, bb)))
    pairs = [(i, j) for i, j in itertools.combinations(bodies, 2) if body_parent.get(i) != j and body_parent.get(j) != i]
    n = a.envs
    cfg = abi.default_config(n, seed=1)
    if a.flags is not None:
        cfg.flags = a.flags
    for kv in a.set:
        k, v = kv.split("="); setattr(cfg, k, float(v))
    o = Oracle(cfg)
    pol = NumpyPolicy()
    rng = np.random.default_rng(1)
    o.step(np.zeros((n, 18), np.float32))
    sep = {p: [] for p in pairs}
    cap = []   # overlap of the capsules the contact model actually uses (tests/scenarios.capsule_penetration), per env-sample
    from tests.scenarios import capsule_penetration
    first = np.ones(n, bool)  # env still in its first episode
    age = np.zeros(n, int)
    for t in range(a.steps):
        obs = o.obs
        if a.policy == "reference":
            act = pol(obs, rng.standard_normal((n, 18)) if a.stochastic else None)
        elif a.policy == "zero":
            act = np.zeros((n, 18), np.float32)
        else:
            act = rng.uniform(-1, 1, (n, 18)).astype(np.float32)
        o.step(act)
        first &= ~(o.reset_buf > 0)
        age += 1
        cap.append(np.where(first & (age > a.skip), capsule_penetration(o, n, model), np.nan))
        rb = o.rigid_body_states.reshape(n, -1, 13).astype(np.float64)
        R = quat_to_mat(rb[:, :, 3:7])
        c, Rm, h = {}, {}, {}
        for b in bodies:
            s = shapes[names[b]]
            c[b] = rb[:, b, 0:3] + np.einsum("nij,j->ni", R[:, b], np.asarray(s["center"]))
            Rm[b] = R[:, b]; h[b] = np.asarray(s["half"])
        for (i, j) in pairs:
            s = obb_separation(c[i], Rm[i], h[i], c[j], Rm[j], h[j])
            sep[(i, j)].append(np.where(first & (age > a.skip), s, np.nan))
    rows = []
    for (i, j), v in sep.items():
        s = np.concatenate(v); s = s[np.isfinite(s)]
        if s.size == 0:
            continue
        rows.append(dict(a=names[i], b=names[j], modelled=(i, j) in modelled, samples=int(s.size), penetrating=float((s < 0).mean()),
                         within_2cm=float((s < 0.02).mean()), deepest=float(max(0.0, -s.min())), median_sep=float(np.median(s))))
    rows.sort(key=lambda r: (-r["penetrating"], -r["within_2cm"]))
    print("policy %s, %d envs x %d steps (first episodes only, the first %d steps after the reset left out), asset %s, flags %s" % (a.policy, n, a.steps, a.skip, a.asset, a.flags))
    print("%-18s %-18s %-9s %11s %11s %9s %10s" % ("body a", "body b", "modelled", "penetrating", "within 2cm", "deepest", "median sep"))
    for r in rows:
        if (r["within_2cm"] > 0.0 and not a.modelled_only) or r["modelled"]:
            print("%-18s %-18s %-9s %10.1f%% %10.1f%% %8.1f mm %8.1f mm" % (r["a"], r["b"], "yes" if r["modelled"] else "NO", 100 * r["penetrating"],
                                                                  100 * r["within_2cm"], 1e3 * r["deepest"], 1e3 * r["median_sep"]))
    cv = np.concatenate(cap); cv = cv[np.isfinite(cv)]
    print("the model's own capsules (what the contact law acts on): overlapping in %.1f%% of the samples, > 5 mm in %.2f%%, p99 %.1f mm, deepest %.1f mm"
          " -- the box columns above add the capsules' geometric error (a capsule's round end leaves up to its radius, 24 mm, of the box's square end uncovered)" % (
              100 * (cv > 0).mean(), 100 * (cv > 0.005).mean(), 1e3 * np.quantile(cv, 0.99), 1e3 * cv.max()))
    never = [r for r in rows if r["within_2cm"] == 0.0 and not r["modelled"]]
    print("%d further unmodelled pairs never come within 2 cm" % len(never))
    if a.out:
        with open(a.out, "w") as f:
            json.dump({"policy": a.policy, "envs": n, "steps": a.steps, "asset": a.asset, "flags": a.flags, "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
