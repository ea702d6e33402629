#!/usr/bin/env python3
"""How well does the in-dynamics joint speed limit hold?  

The speed limit (kick_env.py:327, 2 pi rad/s) is a prescribed-rate joint inside the ABA, decided per joint in pass 2 by a held-parent
predictor of its end-of-substep rate.  A predictor can miss (the joint then exceeds the limit for one substep and is caught by the
next) or fire early (the joint is driven TO the limit).  This tool plays the reference policy / uniform random actions / the leg-press
scenario in the CPU oracle and reports, per control step sample of |qd| / v_lim: share on the limit, share beyond 1.02 / 1.2 / 2, the
maximum -- for the shipped predictor and for the oracle-only exact active-set iteration (tune[22] passes).  Build-container harness.

    python tools/vlimit_probe.py --policy reference --steps 120
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bez_isaacgym_amd import abi  # noqa: E402


def run(policy, n, steps, tune, seed=1, overrides=None):
    from oracle.bez_oracle import Oracle
    from s2s_cpu import NumpyPolicy
    cfg = abi.default_config(n, seed=seed)
    for k, v in (overrides or {}).items():
        setattr(cfg, k, float(v))
    for i, v in tune.items():
        cfg.tune[i] = v
    o = Oracle(cfg)
    pol = NumpyPolicy()
    rng = np.random.default_rng(seed)
    o.step(np.zeros((n, 18), np.float32))
    vl = float(cfg.vel_limit)
    ratios = []
    goals = eps = 0
    ret = np.zeros(n); rets = []
    for t in range(steps):
        obs = o.obs
        if policy == "reference":
            act = pol(obs)
        elif policy == "stochastic":
            act = pol(obs, rng.standard_normal((n, 18)))
        else:
            act = rng.uniform(-1, 1, (n, 18)).astype(np.float32)
        o.step(act)
        r, d = o.rew, o.reset_buf
        ret += r
        for i in np.where(d > 0)[0]:
            eps += 1; goals += int(r[i] > 1.0); rets.append(ret[i]); ret[i] = 0
        qd = np.abs(o.dof_state.reshape(n, 18, 2)[:, 2:, 1].astype(np.float64)) / vl   # head joints are never driven
        ratios.append(qd[d == 0].ravel())
    x = np.concatenate(ratios)
    return dict(samples=int(x.size), on_limit=float(((x > 0.98) & (x <= 1.02)).mean()), over_1_02=float((x > 1.02).mean()),
                over_1_2=float((x > 1.2).mean()), over_2=float((x > 2).mean()), max=float(x.max()), episodes=eps,
                goal_rate=goals / max(eps, 1), mean_return=float(np.mean(rets)) if rets else None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--policy", default="reference", choices=("reference", "stochastic", "random"))
    ap.add_argument("--envs", type=int, default=128)
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--tune", nargs="*", default=[])
    ap.add_argument("--set", nargs="*", default=[])
    a = ap.parse_args()
    tune = {int(kv.split("=")[0]): float(kv.split("=")[1]) for kv in a.tune}
    over = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in a.set}
    print(json.dumps(run(a.policy, a.envs, a.steps, tune, overrides=over)))


if __name__ == "__main__":
    main()
