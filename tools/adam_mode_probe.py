#!/usr/bin/env python3
"""Diagnostic: the optimiser launch (adam_fused_kernel) runs at 9.0 us in some processes and at 12.1 us in others.  Trains a few
epochs as bench.py's PPO leg does and prints where the tensors it touches live, to be set beside the launch's duration in the same process:

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/amp_$i -- python3 tools/adam_mode_probe.py > gpurun_out/amp_$i.log
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bez_isaacgym_amd  # noqa: F401,E402  (before the first HIP call: DESIGN 6.2)
import torch  # noqa: E402
from bez_isaacgym_amd.ppo.a2c_continuous import A2CAgent  # noqa: E402
from bez_isaacgym_amd.utils.config import load_config  # noqa: E402
from bez_isaacgym_amd.utils.rlgames_utils import RLGPUEnv, get_rlgames_env_creator  # noqa: E402

n = 4096
cfg = load_config(["task=bez_kick", "num_envs=%d" % n, "headless=True"])
task = cfg["task"]
task["seed"] = 42
dev = "cuda:0"
task["rl_device"] = dev
venv = RLGPUEnv("rlgpu", n, env_creator=get_rlgames_env_creator(task, "bez_kick", dev, dev, 0, True))
params = cfg["train"]["params"]
params["config"].update(save_frequency=0, save_best_after=10 ** 9)
agent = A2CAgent(params, venv, dev, rank=0, world=1)
agent.obs = agent.env_reset()
for _ in range(int(os.environ.get("EPOCHS", 10))):
    agent.train_epoch()
torch.cuda.synchronize()
names = {"work (ticket)": agent._opt_work, "steps": agent._steps, "lr": agent.lr_t, "loss scale": agent.scaler._scale, "growth tracker": agent.scaler._growth_tracker,
         "grad": agent._flat, "param": agent._pflat, "exp_avg": agent._mflat, "exp_avg_sq": agent._vflat, "param f16": agent._hflat}
for k in ("_norm_parts", "_grid_norm"):
    if getattr(agent, k, None) is not None:
        names[k] = getattr(agent, k)
if agent._packed is not None:
    names.update({"packed": agent._packed.flat, "map_a": agent._packed.map_a, "map_b": agent._packed.map_b})
for k, t in names.items():
    p = t.data_ptr()
    print("PTR %-16s 0x%012x  mod 4K %5d  mod 2M %8d  bytes %d" % (k, p, p % 4096, p % (2 << 20), t.numel() * t.element_size()))
agent.release_env()
