"""Tiny deterministic vectorised env on CPU with the VecTask step()/reset() contract, for PPO tests without a GPU."""
import numpy as np
import torch


class _Box:
    def __init__(self, n):
        self.shape = (n,)


class FakeVecEnv:
    """obs = noisy state in R^54; reward = -|a[:, :3] - tanh(state[:, :3])|^2; episodes of fixed random lengths."""

    def __init__(self, num_envs, seed=0, device="cpu", obs_dim=54, act_dim=18):
        self.n, self.obs_dim, self.act_dim, self.device = num_envs, obs_dim, act_dim, device
        self.gen = torch.Generator().manual_seed(seed)
        self.state = torch.randn(num_envs, obs_dim, generator=self.gen)
        self.t = torch.zeros(num_envs, dtype=torch.long)
        self.horizon = torch.randint(5, 40, (num_envs,), generator=self.gen)
        self.rl_device = device

    def get_env_info(self):
        return {"observation_space": _Box(self.obs_dim), "action_space": _Box(self.act_dim)}

    def reset(self):
        return {"obs": self.state.clone()}

    def step(self, actions):
        a = actions.detach().cpu()
        target = torch.tanh(self.state[:, :3])
        rew = -((a[:, :3] - target) ** 2).sum(-1) * 10.0
        self.state = 0.9 * self.state + 0.1 * torch.randn(self.n, self.obs_dim, generator=self.gen)
        self.t += 1
        done = (self.t >= self.horizon)
        timeout = done.clone()
        if done.any():
            idx = done.nonzero().squeeze(-1)
            self.state[idx] = torch.randn(len(idx), self.obs_dim, generator=self.gen)
            self.t[idx] = 0
        return {"obs": self.state.clone()}, rew, done.long(), {"time_outs": timeout.long()}
