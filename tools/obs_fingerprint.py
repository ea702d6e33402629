#!/usr/bin/env python3
"""The checkpoint's observation statistics as a fingerprint of PhysX.

The reference's shipped checkpoint carries rl_games' obs RunningMeanStd: mean / variance of all 54 observation slots over the
whole training under PhysX (4.03e9 samples; numbers in tests/golden/bez_kick_33_policy.npz).  This tool trains THIS build's PPO in
THIS build's simulator with the reference's hyper-parameters and length for each named variant and prints the agent's own
RunningMeanStd beside the checkpoint's -- per-joint speed sigma, mean pose, IMU and feet-flag rates: which drive / solver variant
leaves the same fingerprint?

    python tools/obs_fingerprint.py --out gpurun_out/r04_fingerprint.json --epochs 6156 --seeds 42 \
        --run default: --run kd2:task.env.control.damping=2
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FIXTURE = os.path.join(ROOT, "tests", "golden", "bez_kick_33_policy.npz")
GROUPS = (("q head", 0, 2), ("q arms", (2, 3, 10, 11)), ("q legs", (4, 5, 6, 7, 8, 9, 12, 13, 14, 15, 16, 17)), ("qd head", 18, 20),
          ("qd arms", (20, 21, 28, 29)), ("qd legs", (22, 23, 24, 25, 26, 27, 30, 31, 32, 33, 34, 35)), ("imu", 36, 42), ("off_orn", 42, 44),
          ("feet", 44, 52))


def reference():
    d = np.load(FIXTURE)
    return d["running_mean_std/running_mean"].astype(np.float64), d["running_mean_std/running_var"].astype(np.float64)


def train_one(overrides, seed, epochs):
    import torch
    from bez_isaacgym_amd.ppo.a2c_continuous import A2CAgent
    from bez_isaacgym_amd.utils.config import load_config
    from bez_isaacgym_amd.utils.rlgames_utils import RLGPUEnv, get_rlgames_env_creator
    from bez_isaacgym_amd.utils.utils import set_seed
    cfg = load_config(["task=bez_kick", "num_envs=4096", "headless=True", "seed=%d" % seed] + list(overrides))
    set_seed(seed)
    task = cfg["task"]
    task["seed"] = seed
    dev = "cuda:0"
    venv = RLGPUEnv("rlgpu", 4096, env_creator=get_rlgames_env_creator(task, "bez_kick", dev, dev, 0, True))
    params = cfg["train"]["params"]
    params["config"].update(save_frequency=0, save_best_after=10 ** 9)
    agent = A2CAgent(params, venv, dev)
    rewards = []
    agent.train(max_epochs=epochs, log=lambda s: rewards.append(float(s.rsplit("mean_reward", 1)[1])) if "mean_reward" in s else None)
    torch.cuda.synchronize()
    rms = agent.running_mean_std
    tail = [r for r in rewards[-2000:] if r == r]
    return (rms.running_mean.cpu().numpy().copy(), rms.running_var.cpu().numpy().copy(), float(rms.count.item()),
            float(np.mean(tail)) if tail else float("nan"))


def table(name, mean, var, ref_mean, ref_var, out=print):
    sd, rsd = np.sqrt(var), np.sqrt(ref_var)
    z = (mean - ref_mean) / np.sqrt(ref_var + 1e-5)
    out("  %-22s  qd sigma arms %s | legs L %s | legs R %s" % (name, np.round(sd[[20, 21, 28, 29]], 2), np.round(sd[22:28], 2), np.round(sd[30:36], 2)))
    out("  %-22s  feet mean %s  imu sigma %s" % ("", np.round(mean[44:52], 2), np.round(sd[36:42], 3)))
    out("  %-22s  rms of z(mean) per group / rms of log(sigma ratio) per group:" % "")
    res = {}
    for g in GROUPS:
        idx = np.arange(g[1], g[2]) if len(g) == 3 else np.array(g[1])
        zr = float(np.sqrt(np.mean(z[idx] ** 2)))
        lr = float(np.sqrt(np.mean(np.log((sd[idx] + 1e-3) / (rsd[idx] + 1e-3)) ** 2)))
        res[g[0]] = (zr, lr)
        out("      %-8s z %.2f   log-sigma %.2f" % (g[0], zr, lr))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/obs_fingerprint.json")
    ap.add_argument("--seeds", type=int, nargs="+", default=[42])
    ap.add_argument("--epochs", type=int, default=6156)
    ap.add_argument("--run", action="append", default=[])
    a = ap.parse_args()
    rm, rv = reference()
    print("reference checkpoint (PhysX):")
    table("Bez_Kick_33.pth", rm, rv, rm, rv)
    results = {"reference": {"mean": rm.tolist(), "var": rv.tolist()}}
    for spec in (a.run or ["default:"]):
        tag, _, ov = spec.partition(":")
        for seed in a.seeds:
            t0 = time.time()
            mean, var, count, rew = train_one(ov.split(), seed, a.epochs)
            name = "%s_s%d" % (tag, seed)
            print("%s: %d epochs, last-2000-epoch mean reward %.2f, %.3g samples in the normaliser (%.0f s)" % (name, a.epochs, rew, count, time.time() - t0), flush=True)
            g = table(name, mean, var, rm, rv)
            results[name] = {"mean": mean.tolist(), "var": var.tolist(), "count": count, "reward": rew, "groups": g, "overrides": ov}
            with open(a.out, "w") as f:
                json.dump(results, f)


if __name__ == "__main__":
    main()
