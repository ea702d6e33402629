#!/usr/bin/env python3
"""Sim-to-sim check of the REFERENCE policy in the HIP simulator, and a sweep of the simulator's free physics
parameters against it.

The reference ships one trained policy (results/Bez_Kick/Normal/Bez_Kick_33.pth; numeric fixture
tests/golden/bez_kick_33_policy.npz).  It is played deterministically (mu, clamp +-1: utils/players.py:46-66) in N
parallel envs through the C ABI; the report is what a PhysX-faithful simulator would have to reproduce: goal rate,
mean episode return (the checkpoint's last_mean_rewards = 87.55), episode length, termination reasons, and the
per-dimension distance between the rollout's observation statistics and the checkpoint's own running mean / var.

    python tools/sim2sim_gpu.py                      # one evaluation with the default BezSimConfig
    python tools/sim2sim_gpu.py --set contact_kn=4e4 contact_cn=80
    python tools/sim2sim_gpu.py --sweep 200          # random search (log-uniform around the defaults), best first
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bez_isaacgym_amd import abi  # noqa: E402
from bez_isaacgym_amd.sim import BezSim  # noqa: E402
from bez_isaacgym_amd.utils.player import PpoPlayerContinuous  # noqa: E402

FIXTURE = os.path.join(ROOT, "tests", "golden", "bez_kick_33_policy.npz")

# knob -> (low, high, log scale?) : the fields of BezSimConfig that PhysX gives no value for
KNOBS = {
    "contact_kn": (3e3, 2e5, True), "contact_cn": (5.0, 2e3, True), "contact_ct": (50.0, 2e4, True),
    "contact_veps": (1e-3, 0.1, True), "jfric_veps": (0.01, 1.0, True), "limit_k": (50.0, 5e3, True),
    "limit_d": (0.5, 50.0, True), "ball_ang_damping": (0.0, 1.0, False), "self_kn": (5e2, 3e4, True),
    "self_cn": (1.0, 50.0, True), "joint_friction": (0.0, 0.3, False), "plane_friction": (0.6, 1.4, False),
}


def evaluate(player, overrides=None, n=4096, steps=900, seed=1, flags=None, device=0, collect_obs=True):
    """Play the policy for `steps` control steps in n envs; episodes that finish inside the window are scored."""
    cfg = abi.default_config(n, seed=seed)
    for k, v in (overrides or {}).items():
        setattr(cfg, k, type(getattr(cfg, k))(v))
    if flags is not None:
        cfg.flags = flags
    sim = BezSim(cfg, device)
    dev = sim.device
    obs = sim.tensor(abi.TENSOR_OBS); rew = sim.tensor(abi.TENSOR_REW); rst = sim.tensor(abi.TENSOR_RESET)
    prog = sim.tensor(abi.TENSOR_PROGRESS)
    sim.step(torch.zeros(n * 18, device=dev))
    ret = torch.zeros(n, device=dev); length = torch.zeros(n, device=dev)
    acc = dict(episodes=0, ret=0.0, len=0.0, goal=0, timeout=0, fall=0, oob=0, angle=0, goal_len=0.0)
    osum = torch.zeros(54, device=dev, dtype=torch.float64); osq = torch.zeros(54, device=dev, dtype=torch.float64); ocount = 0
    miss_y, miss_v = [], []  # where / how fast the ball passes the goal line when an episode ends by the goal-angle test
    for t in range(steps):
        a = player.get_action(obs)
        if collect_obs:
            o64 = obs.double(); osum += o64.sum(0); osq += (o64 * o64).sum(0); ocount += n
        sim.step(a.reshape(-1).contiguous())
        ret += rew; length += 1
        done = rst > 0
        if bool(done.any()):
            root = sim.refresh(abi.TENSOR_ROOT_STATE).view(n, 2, 13)
            d = done.nonzero().squeeze(-1)
            r, z = rew[d], root[d, 0, 2]
            xy = torch.linalg.norm(root[d, 0, :2] - torch.tensor(list(cfg.bez_init[:2]), device=dev), dim=1)
            goal = r > 1.0
            tmo = (~goal) & (prog[d] >= cfg.max_episode_length)
            fall = (~goal) & (~tmo) & (z < 0.275)
            oob = (~goal) & (~tmo) & (~fall) & (xy > 0.5)
            acc["episodes"] += int(d.numel()); acc["ret"] += float(ret[d].sum()); acc["len"] += float(length[d].sum())
            acc["goal"] += int(goal.sum()); acc["timeout"] += int(tmo.sum()); acc["fall"] += int(fall.sum()); acc["oob"] += int(oob.sum())
            acc["angle"] += int(d.numel()) - int(goal.sum() + tmo.sum() + fall.sum() + oob.sum())
            acc["goal_len"] += float(length[d][goal].sum())
            ang = (~goal) & (~tmo) & (~fall) & (~oob)
            if bool(ang.any()):
                miss_y.append(root[d, 1, 1][ang].cpu()); miss_v.append(torch.linalg.norm(root[d, 1, 7:9][ang], dim=1).cpu())
            ret[d] = 0; length[d] = 0
    torch.cuda.synchronize()
    e = max(acc["episodes"], 1)
    out = dict(episodes=acc["episodes"], goal_rate=acc["goal"] / e, mean_return=acc["ret"] / e, mean_length=acc["len"] / e,
               goal_length=acc["goal_len"] / max(acc["goal"], 1),
               reasons={k: acc[k] for k in ("goal", "fall", "oob", "angle", "timeout")})
    if miss_y:
        my, mv = torch.cat(miss_y).abs().numpy(), torch.cat(miss_v).numpy()
        out["angle_miss_abs_y_m"] = {"p10": float(np.percentile(my, 10)), "p50": float(np.percentile(my, 50)), "p90": float(np.percentile(my, 90))}
        out["angle_miss_ball_speed"] = {"p10": float(np.percentile(mv, 10)), "p50": float(np.percentile(mv, 50)), "p90": float(np.percentile(mv, 90))}
    if collect_obs:
        ck = player.checkpoint["running_mean_std"]
        rm, rv = np.asarray(ck["running_mean"], np.float64), np.asarray(ck["running_var"], np.float64)
        mean = (osum / ocount).cpu().numpy(); var = (osq / ocount).cpu().numpy() - mean * mean
        z = (mean - rm) / np.sqrt(rv + 1e-5)
        out["obs_z_rms"] = float(np.sqrt(np.mean(z[:52] ** 2)))
        out["obs_z"] = [round(float(v), 2) for v in z]
        out["obs_std_ratio"] = [round(float(v), 2) for v in np.sqrt(np.maximum(var, 0) / (rv + 1e-5))]
    sim.close()
    return out


def sample(rng):
    o = {}
    for k, (lo, hi, lg) in KNOBS.items():
        o[k] = math.exp(rng.uniform(math.log(lo), math.log(hi))) if lg else rng.uniform(lo, hi)
    return o


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=900)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--set", nargs="*", default=[], help="field=value overrides of BezSimConfig")
    ap.add_argument("--flags", type=int, default=None)
    ap.add_argument("--sweep", type=int, default=0, help="number of random parameter draws")
    ap.add_argument("--sweep-steps", type=int, default=300)
    ap.add_argument("--sweep-envs", type=int, default=1024)
    ap.add_argument("--checkpoint", default=FIXTURE)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    player = PpoPlayerContinuous(a.checkpoint, "cuda:0")
    over = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in a.set}
    t0 = time.time()
    if a.sweep:
        rng = np.random.default_rng(a.seed)
        rows = [dict(over={}, **evaluate(player, over, a.sweep_envs, a.sweep_steps, a.seed, a.flags, collect_obs=False))]
        for i in range(a.sweep):
            o = dict(sample(rng), **over)
            r = evaluate(player, o, a.sweep_envs, a.sweep_steps, a.seed, a.flags, collect_obs=False)
            rows.append(dict(over=o, **r))
            if i % 20 == 19:
                print("[sweep] %d/%d  best so far mean_length %.1f  (%.0f s)" % (i + 1, a.sweep, max(x["mean_length"] for x in rows), time.time() - t0), flush=True)
        rows.sort(key=lambda x: (-x["goal_rate"], -x["mean_length"]))
        res = dict(kind="sweep", rows=rows[:25], default=[x for x in rows if not x["over"]][0])
    else:
        res = dict(kind="eval", overrides=over, **evaluate(player, over, a.envs, a.steps, a.seed, a.flags))
    res["seconds"] = time.time() - t0
    txt = json.dumps(res, indent=1)
    print(txt)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt)


if __name__ == "__main__":
    main()
