#!/usr/bin/env python3
"""Sim-to-sim of the REFERENCE policy in the CPU ORACLE (build-container experiment harness; never on the product path).

The reference's shipped checkpoint (numeric fixture tests/golden/bez_kick_33_policy.npz, made by
bez_isaacgym_amd/utils/rlg_checkpoint.py without unpickling) is played deterministically (mu, clamp +-1:
utils/players.py:46-66 of the reference) in N oracle envs.  Reported: goal rate, mean return (the checkpoint's
last_mean_rewards is 87.55), episode length, termination reasons and the z-distance of the rollout's observation
statistics to the checkpoint's running mean / var.  Used to judge physics-model variants before they go into the HIP
kernels.

    python tools/s2s_cpu.py                          # defaults
    python tools/s2s_cpu.py --set contact_cn=155     # BezSimConfig overrides
    python tools/s2s_cpu.py --flags 65               # flag word
    python tools/s2s_cpu.py --trace 3                # time series of env 3's first episode
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bez_isaacgym_amd import abi  # noqa: E402
from oracle.bez_oracle import Oracle  # noqa: E402

FIXTURE = os.path.join(ROOT, "tests", "golden", "bez_kick_33_policy.npz")


class NumpyPolicy:
    """actor MLP 54-400-200-100 (ELU) + mu head, obs RunningMeanStd with clamp +-5 (rl_games [ext])."""

    def __init__(self, path=FIXTURE):
        d = np.load(path)
        g = lambda k: d["model/a2c_network." + k].astype(np.float64)
        self.W = [g("actor_mlp.%d.weight" % i) for i in (0, 2, 4)]
        self.b = [g("actor_mlp.%d.bias" % i) for i in (0, 2, 4)]
        self.Wmu, self.bmu = g("mu.weight"), g("mu.bias")
        self.Wv, self.bv = g("value.weight"), g("value.bias")
        self.mean = d["running_mean_std/running_mean"].astype(np.float64)
        self.var = d["running_mean_std/running_var"].astype(np.float64)
        self.vmean = float(d["reward_mean_std/running_mean"][0]); self.vvar = float(d["reward_mean_std/running_var"][0])
        self.logstd = g("sigma")

    def features(self, obs):
        x = np.clip((obs.astype(np.float64) - self.mean) / np.sqrt(self.var + 1e-5), -5.0, 5.0)
        for W, b in zip(self.W, self.b):
            x = x @ W.T + b
            x = np.where(x > 0, x, np.expm1(np.minimum(x, 0)))
        return x

    def __call__(self, obs, noise=None):
        x = self.features(obs)
        mu = x @ self.Wmu.T + self.bmu
        if noise is not None:
            mu = mu + np.exp(self.logstd) * noise
        return np.clip(mu, -1.0, 1.0).astype(np.float32)

    def value(self, obs):
        v = self.features(obs) @ self.Wv.T + self.bv
        return v[:, 0] * np.sqrt(self.vvar + 1e-5) + self.vmean


def make_cfg(n, seed, overrides, flags, tune):
    cfg = abi.default_config(n, seed=seed)
    for k, v in (overrides or {}).items():
        setattr(cfg, k, type(getattr(cfg, k))(v))
    if flags is not None:
        cfg.flags = flags
    for i, v in (tune or {}).items():
        cfg.tune[int(i)] = float(v)
    return cfg


def evaluate(policy, overrides=None, flags=None, tune=None, n=256, steps=600, seed=1, stochastic=False, precision="f64"):
    cfg = make_cfg(n, seed, overrides, flags, tune)
    o = Oracle(cfg, precision=precision)
    o.step(np.zeros((n, 18), np.float32))
    rng = np.random.default_rng(seed)
    ret = np.zeros(n); length = np.zeros(n)
    acc = dict(episodes=0, ret=0.0, len=0.0, goal=0, timeout=0, fall=0, oob=0, angle=0, goal_len=0.0)
    osum = np.zeros(54); osq = np.zeros(54); ocount = 0
    binit = np.array(list(cfg.bez_init[:2]))
    for t in range(steps):
        obs = o.obs
        osum += obs.sum(0, dtype=np.float64); osq += (obs.astype(np.float64) ** 2).sum(0); ocount += n
        a = policy(obs, rng.standard_normal((n, 18)) if stochastic else None)
        o.step(a)
        r, d = o.rew, o.reset_buf
        ret += r; length += 1
        idx = np.where(d > 0)[0]
        if idx.size:
            root = o.root_states.reshape(n, 2, 13)
            prog = o.progress_buf
            for i in idx:
                acc["episodes"] += 1; acc["ret"] += ret[i]; acc["len"] += length[i]
                if r[i] > 1.0:
                    acc["goal"] += 1; acc["goal_len"] += length[i]
                elif prog[i] >= cfg.max_episode_length:
                    acc["timeout"] += 1
                elif root[i, 0, 2] < 0.275:
                    acc["fall"] += 1
                elif np.linalg.norm(root[i, 0, :2] - binit) > 0.5:
                    acc["oob"] += 1
                else:
                    acc["angle"] += 1
                ret[i] = 0; length[i] = 0
    e = max(acc["episodes"], 1)
    mean = osum / ocount; var = osq / ocount - mean * mean
    z = (mean - policy.mean) / np.sqrt(policy.var + 1e-5)
    return dict(episodes=acc["episodes"], goal_rate=acc["goal"] / e, mean_return=acc["ret"] / e, mean_length=acc["len"] / e,
                goal_length=acc["goal_len"] / max(acc["goal"], 1),
                reasons={k: acc[k] for k in ("goal", "fall", "oob", "angle", "timeout")},
                obs_z_rms=float(np.sqrt(np.mean(z[:52] ** 2))), obs_z=[round(float(v), 2) for v in z],
                obs_std_ratio=[round(float(v), 2) for v in np.sqrt(np.maximum(var, 0) / (policy.var + 1e-5))])


def trace(policy, env, overrides=None, flags=None, tune=None, n=64, steps=200, seed=1):
    """Time series of one env until its first reset."""
    cfg = make_cfg(n, seed, overrides, flags, tune)
    o = Oracle(cfg)
    o.step(np.zeros((n, 18), np.float32))
    np.set_printoptions(precision=3, suppress=True, linewidth=250)
    print("  t     z    roll  pitch   yaw |  x      y   | ball x  y   vx   |  Lfz   Rfz | act L-leg (6)                       | act R-leg (6)                  | value")
    for t in range(steps):
        obs = o.obs
        a = policy(obs)
        v = policy.value(obs)
        rs = o.root_states.reshape(n, 2, 13)[env]
        x, y, zq, w = rs[0, 3:7]
        roll = np.arctan2(2 * (w * x + y * zq), 1 - 2 * (x * x + y * y))
        pitch = np.arcsin(np.clip(2 * (w * y - zq * x), -1, 1))
        yaw = np.arctan2(2 * (w * zq + x * y), 1 - 2 * (y * y + zq * zq))
        cf = o.contact_forces.reshape(n, -1, 3)[env]
        print("%3d %6.3f %6.3f %6.3f %6.3f | %6.3f %6.3f | %5.3f %6.3f %5.2f | %5.1f %5.1f | %s | %s | %6.1f" % (
            t, rs[0, 2], roll, pitch, yaw, rs[0, 0], rs[0, 1], rs[1, 0], rs[1, 1], rs[1, 7], cf[12, 2], cf[20, 2],
            np.array2string(a[env, 4:10]), np.array2string(a[env, 12:18]), v[env]))
        o.step(a)
        if o.reset_buf[env] > 0:
            print("reset at t=%d rew %.2f" % (t, o.rew[env]))
            break


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=256)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--set", nargs="*", default=[])
    ap.add_argument("--tune", nargs="*", default=[], help="index=value entries of BezSimConfig.tune")
    ap.add_argument("--flags", type=int, default=None)
    ap.add_argument("--stochastic", action="store_true")
    ap.add_argument("--trace", type=int, default=None)
    ap.add_argument("--precision", default="f64")
    ap.add_argument("--brief", action="store_true")
    a = ap.parse_args()
    pol = NumpyPolicy()
    over = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in a.set}
    tune = {int(kv.split("=")[0]): float(kv.split("=")[1]) for kv in a.tune}
    if a.trace is not None:
        trace(pol, a.trace, over, a.flags, tune, seed=a.seed)
        return
    t0 = time.time()
    r = evaluate(pol, over, a.flags, tune, a.envs, a.steps, a.seed, a.stochastic, a.precision)
    r["seconds"] = round(time.time() - t0, 1)
    if a.brief:
        z = r["obs_z"]
        print("goal %.3f ret %7.2f len %6.1f glen %5.1f %s zrms %.2f z43 %.2f z42 %.2f z38 %.2f z3 %.2f" % (
            r["goal_rate"], r["mean_return"], r["mean_length"], r["goal_length"], r["reasons"], r["obs_z_rms"], z[43], z[42], z[38], z[3]))
    else:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
