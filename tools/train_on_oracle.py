#!/usr/bin/env python3
"""EXPERIMENT harness (never on the product path): train this build's PPO loop (GPU) on the CPU ORACLE as the environment, to
judge a physics variant that exists only in the oracle -- the rigid-contact model (BEZ_FLAG_HARD_CONTACT: PGS with stiction and
restitution 0) -- by what training under it reaches, before any of it is ported to the HIP kernels (VERDICT round 2, item 1 (c),(d)).

The oracle steps N envs on the host cores (OpenMP); observations / rewards / resets are copied into persistent device buffers, so
the agent's rollout kernels see the same contract the HIP env gives them.  HIP graphs are off (the env is a host call).

    python tools/train_on_oracle.py --flags 65 --epochs 1500 --seed 42 [--tune 7=1] [--set ball_cn=155] [--out gpurun_out/x.json]
"""
import argparse
import json
import os
import sys
import time

# libgomp reads this when the oracle library is first loaded: a 1-GPU box's CPU share is 16 cores, more threads than that only spin
os.environ.setdefault("OMP_NUM_THREADS", str(min(len(os.sched_getaffinity(0)), 16)))

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bez_isaacgym_amd import abi  # noqa: E402
from oracle.bez_oracle import Oracle  # noqa: E402


class _Box:
    def __init__(self, n):
        self.shape = (n,)


class OracleVecEnv:
    """VecTask step()/reset() contract over oracle.bez_oracle.Oracle; tensors live on `device`."""

    def __init__(self, task_cfg, device, flags=None, tune=None, overrides=None, precision="f64"):
        seed = int(task_cfg.get("seed", 42))
        cfg = abi.config_from_task_cfg(task_cfg, seed=seed, env_id_offset=0, strict_reference_quirks=True, task="bez_kick")
        if flags is not None:
            cfg.flags = int(flags)
        for k, v in (overrides or {}).items():
            setattr(cfg, k, type(getattr(cfg, k))(v))
        for i, v in (tune or {}).items():
            cfg.tune[int(i)] = float(v)
        self.sim_cfg = cfg
        self.o = Oracle(cfg, precision=precision)
        self.n = int(cfg.num_envs)
        self.num_envs = self.n
        self.rl_device = device
        self.device = torch.device(device)
        self.env = self
        self.observation_space, self.action_space = _Box(54), _Box(18)
        d = self.device
        self.obs_buf = torch.zeros(self.n, 54, device=d)
        self.rew_buf = torch.zeros(self.n, device=d)
        self.reset_buf = torch.zeros(self.n, device=d, dtype=torch.int64)
        self.timeout_buf = torch.zeros(self.n, device=d, dtype=torch.int64)
        self.extras = {"time_outs": self.timeout_buf}
        self.env_seconds = 0.0
        self.stats = dict(episodes=0, goal=0, ret=0.0, length=0.0)
        self._ret = np.zeros(self.n); self._len = np.zeros(self.n)

    def get_env_info(self):
        return {"observation_space": self.observation_space, "action_space": self.action_space}

    def _publish(self):
        o = self.o
        self.obs_buf.copy_(torch.from_numpy(np.ascontiguousarray(o.obs, dtype=np.float32)))
        self.rew_buf.copy_(torch.from_numpy(np.ascontiguousarray(o.rew, dtype=np.float32)))
        self.reset_buf.copy_(torch.from_numpy(np.ascontiguousarray(o.reset_buf, dtype=np.int64)))
        self.timeout_buf.copy_(torch.from_numpy(np.ascontiguousarray(o.timeout_buf, dtype=np.int64)))

    def step(self, actions):
        a = actions.detach().to("cpu", torch.float32).numpy().reshape(self.n, 18)
        t0 = time.perf_counter()
        self.o.step(np.ascontiguousarray(a))
        self.env_seconds += time.perf_counter() - t0
        r = np.asarray(self.o.rew, dtype=np.float64); done = np.asarray(self.o.reset_buf) > 0
        self._ret += r; self._len += 1
        if done.any():
            s = self.stats
            s["episodes"] += int(done.sum()); s["goal"] += int((done & (r > 5.0)).sum())
            s["ret"] += float(self._ret[done].sum()); s["length"] += float(self._len[done].sum())
            self._ret[done] = 0; self._len[done] = 0
        self._publish()
        return {"obs": self.obs_buf}, self.rew_buf, self.reset_buf, self.extras

    def reset(self):
        self.step(torch.zeros(self.n, 18))
        return {"obs": self.obs_buf}

    def pop_stats(self):
        s, self.stats = self.stats, dict(episodes=0, goal=0, ret=0.0, length=0.0)
        e = max(s["episodes"], 1)
        return dict(episodes=s["episodes"], goal_rate=s["goal"] / e, mean_return=s["ret"] / e, mean_length=s["length"] / e)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", type=int, default=65)
    ap.add_argument("--tune", nargs="*", default=[], help="slot=value of BezSimConfig.tune")
    ap.add_argument("--set", nargs="*", default=[], help="field=value of BezSimConfig")
    ap.add_argument("--epochs", type=int, default=1500)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--every", type=int, default=50)
    ap.add_argument("--precision", default="f64")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--out", default="gpurun_out/r03_train_on_oracle.json")
    ap.add_argument("overrides", nargs="*", help="extra hydra-style overrides for the PPO config")
    a = ap.parse_args()
    from bez_isaacgym_amd.ppo.a2c_continuous import A2CAgent
    from bez_isaacgym_amd.utils.config import load_config
    from bez_isaacgym_amd.utils.utils import set_seed
    cfg = load_config(["task=bez_kick", "num_envs=%d" % a.envs, "headless=True", "seed=%d" % a.seed, "max_iterations=%d" % a.epochs,
                       "train.params.config.hip_graphs=False", "train.params.config.lean_env_step=False",
                       "train.params.config.save_frequency=0"] + list(a.overrides))
    cfg["seed"] = set_seed(cfg["seed"])
    task_cfg = cfg["task"]; task_cfg["seed"] = cfg["seed"]
    tune = {int(k): float(v) for k, v in (s.split("=") for s in a.tune)}
    over = {k: float(v) for k, v in (s.split("=") for s in a.set)}
    env = OracleVecEnv(task_cfg, a.device, flags=a.flags, tune=tune, overrides=over, precision=a.precision)
    agent = A2CAgent(cfg["train"]["params"], env, a.device)
    agent.obs = agent.env_reset()
    curve, t0 = [], time.time()
    for ep in range(1, a.epochs + 1):
        st = agent.train_epoch()
        if ep % a.every == 0 or ep == a.epochs:
            s = env.pop_stats()
            row = dict(epoch=ep, frames=ep * agent.batch_size, mean_reward=(agent.game_rewards[0] if agent.game_rewards else float("nan")),
                       window=s, lr=st["lr"], kl=st["kl"], seconds=time.time() - t0, env_seconds=env.env_seconds)
            curve.append(row)
            print("epoch %d frames %d mean_reward %.2f | last %d epochs: goal_rate %.3f return %.2f length %.1f | lr %.2e | %.0f s (env %.0f s)" % (
                ep, row["frames"], row["mean_reward"], a.every, s["goal_rate"], s["mean_return"], s["mean_length"], st["lr"], row["seconds"], env.env_seconds), flush=True)
            with open(a.out, "w") as f:
                json.dump(dict(args=vars(a), curve=curve), f, indent=1)
    sigma = torch.exp(agent.model.a2c_network.sigma.detach()).cpu().numpy().round(3).tolist()
    print("sigma", sigma)
    with open(a.out, "w") as f:
        json.dump(dict(args=vars(a), curve=curve, sigma=sigma), f, indent=1)


if __name__ == "__main__":
    main()
