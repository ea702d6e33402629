"""rl_games-side adapter with the reference's surface (bez_isaacgym/utils/rlgames_utils.py): the env factory
get_rlgames_env_creator (:39-98), RLGPUEnv (:157-181) and the stats observer (:101-154).  rl_games itself is not a
dependency: the consumer is bez_isaacgym_amd.ppo.  Multi-GPU follows the reference's Horovod scheme (rank -> cuda:{rank},
:71-81) with torch.distributed ranks instead."""
import csv
import json
import os
import time
from typing import Callable

import torch

from ..tasks import isaacgym_task_map


def get_rlgames_env_creator(task_config: dict, task_name: str, sim_device: str, rl_device: str, graphics_device_id: int,
                            headless: bool, multi_gpu: bool = False, post_create_hook: Callable = None):
    def create_rlgpu_env(_sim_device=sim_device, _rl_device=rl_device, **kwargs):
        if multi_gpu:
            rank = int(os.environ.get("RANK", "0"))
            local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
            print("Rank: ", rank)
            _sim_device = f"cuda:{local_rank}"
            _rl_device = f"cuda:{local_rank}"
            task_config["rank"] = rank
            task_config["rl_device"] = _rl_device
            # global env ids: reset noise must not depend on how many GPUs share the job
            task_config["env_id_offset"] = rank * int(task_config["env"]["numEnvs"])
        else:
            _sim_device, _rl_device = sim_device, rl_device
            task_config.setdefault("rl_device", rl_device)
        env = isaacgym_task_map[task_name](cfg=task_config, sim_device=_sim_device,
                                           graphics_device_id=graphics_device_id, headless=headless)
        if post_create_hook is not None:
            post_create_hook()
        return env
    return create_rlgpu_env


class RLGPUEnv:
    """vecenv.IVecEnv-shaped wrapper (rlgames_utils.py:157-181)."""

    def __init__(self, config_name, num_actors, env_creator=None, **kwargs):
        self.env = env_creator(**kwargs)

    def step(self, action):
        return self.env.step(action)

    def reset(self):
        return self.env.reset()

    def get_number_of_agents(self):
        return self.env.get_number_of_agents()

    def get_env_info(self):
        info = {"action_space": self.env.action_space, "observation_space": self.env.observation_space}
        if self.env.num_states > 0:
            info["state_space"] = self.env.state_space
        return info


class RLGPUAlgoObserver:
    """Scalar logging with the reference's scalar names (rlgames_utils.py:137-154).  TensorBoard is not installed here,
    so scalars go to runs/<name>/scalars.jsonl (+ .csv) -- same tags: scores/mean, scores/iter, scores/time, ..."""

    def __init__(self, run_dir):
        os.makedirs(run_dir, exist_ok=True)
        self.path = os.path.join(run_dir, "scalars.jsonl")
        self.t0 = time.time()
        self.f = open(self.path, "a")

    def add(self, row):
        mean = row.get("mean_reward")
        out = dict(row)
        if mean is not None and mean == mean:
            # rlgames_utils.py:150-153: add_scalar('scores/mean', m, frame), ('scores/iter', m, epoch_num), ('scores/time', m, total_time) --
            # one value under three tags that differ by their STEP axis; a line-oriented log keeps (value, step) per tag
            out["scalars"] = {"scores/mean": [mean, row.get("frame")], "scores/iter": [mean, row.get("epoch")], "scores/time": [mean, row.get("time")]}
        self.f.write(json.dumps(out) + "\n")
        self.f.flush()
