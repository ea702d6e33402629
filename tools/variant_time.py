#!/usr/bin/env python3
"""Fused-step time of the kernel variants on one box: default asset, per-env (domain-randomisation) parameters, cleats asset, bez_walk."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, numpy as np
from bez_isaacgym_amd import abi
from bez_isaacgym_amd.sim import BezSim
def run(cfg, dr):
    sim = BezSim(cfg, 0)
    n = cfg.num_envs
    if dr:
        sim.set_env_params(abi.PARAM_FRICTION, torch.ones(n, device="cuda"))
        sim.set_env_params(abi.PARAM_KP_SCALE, torch.ones(n * 18, device="cuda"))
        sim.set_env_params(abi.PARAM_MASS_SCALE, torch.ones(n * 19, device="cuda"))
    acts = (torch.rand(64, n * 18, device="cuda") * 2 - 1).contiguous()
    for t in range(100): sim.step(acts[t % 64])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(1500): sim.step(acts[t % 64])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 1500
c = abi.default_config(4096)
print("default      %.2f us" % run(c, False))
print("DR variant   %.2f us" % run(abi.default_config(4096), True))
c = abi.default_config(4096); c.flags |= abi.FLAG_CLEATS
print("cleats       %.2f us" % run(c, False))
c = abi.default_config(4096); c.task = abi.TASK_IDS["bez_walk"]; c.max_episode_length = 600
print("walk         %.2f us" % run(c, False))
