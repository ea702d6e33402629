"""Entry point with the reference's CLI contract (bez_isaacgym/train.py:60-116):

    python -m bez_isaacgym_amd.train task=bez_kick num_envs=4096 sim_device=cuda:0 pipeline=gpu headless=True
    python -m torch.distributed.run --nproc-per-node 8 -m bez_isaacgym_amd.train task=bez_kick multi_gpu=True headless=True

Composes cfg/config.yaml + task + train groups (utils/config.py), seeds, builds the env through the rl_games-style
factory, dumps runs/<name>/config.yaml (train.py:105-108) and runs the PPO loop (ppo/a2c_continuous.py).
"""
import os
import sys

import yaml


def play(cfg, vec_env, params, games=None, max_steps=None, log=print):
    """test=True: the reference's play loop (play.py:125-236, utils/players.py:46-72) -- restore a checkpoint into the
    deterministic continuous player and run whole episodes in all envs; returns (mean reward, mean steps, games)."""
    import torch
    from .utils.player import PpoPlayerContinuous
    env = vec_env.env
    path = params.get("load_path")
    if not path:
        raise ValueError("test=True needs checkpoint=<file> (an rl_games / this build's .pth, or the numeric .npz fixture)")
    obs_dim, act_dim = env.observation_space.shape[0], env.action_space.shape[0]
    player = PpoPlayerContinuous(path, env.rl_device, obs_dim, act_dim, tuple(params["network"]["mlp"]["units"]))
    pc = params["config"].get("player", {}) or {}
    n_games = int(games or pc.get("games_num", 2000))                    # rl_games BasePlayer defaults [ext]
    max_steps = int(max_steps or pc.get("max_steps", 27000 // 4))
    obs = env.reset()["obs"]
    n = env.num_envs
    cr, steps = torch.zeros(n, device=env.rl_device), torch.zeros(n, device=env.rl_device)
    sum_r = sum_s = 0.0
    played = 0
    for _ in range(max_steps):
        obs_d, r, done, _info = env.step(player.get_action(obs))
        obs = obs_d["obs"]
        cr += r; steps += 1
        idx = done.nonzero(as_tuple=False).squeeze(-1)
        if idx.numel():
            sum_r += float(cr[idx].sum()); sum_s += float(steps[idx].sum()); played += int(idx.numel())
            cr[idx] = 0; steps[idx] = 0
            if played >= n_games:
                break
    mean_r, mean_s = sum_r / max(played, 1), sum_s / max(played, 1)
    log("av reward: %.3f av steps: %.1f games: %d" % (mean_r, mean_s, played))  # play.py:227-231 wording
    return mean_r, mean_s, played


def launch(argv=None, vec_env_factory=None):
    """`vec_env_factory(task_cfg, rank)` replaces the HIP env (tests drive the whole CLI / multi-GPU wiring on CPU with it)."""
    import torch
    import torch.distributed as dist
    from .ppo.a2c_continuous import A2CAgent
    from .utils.config import load_config, print_dict
    from .utils.rlgames_utils import RLGPUAlgoObserver, RLGPUEnv, get_rlgames_env_creator
    from .utils.utils import set_np_formatting, set_seed

    cfg = load_config(list(sys.argv[1:] if argv is None else argv))
    if cfg.get("checkpoint"):
        cfg["checkpoint"] = os.path.abspath(cfg["checkpoint"])
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    multi_gpu = bool(cfg.get("multi_gpu", False)) and world > 1
    if multi_gpu and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # RCCL over xGMI
        else:
            dist.init_process_group("gloo")
    if rank == 0:
        print_dict(cfg)
    set_np_formatting()
    cfg["seed"] = set_seed(cfg["seed"], torch_deterministic=cfg.get("torch_deterministic", False))
    task_cfg = cfg["task"]
    task_cfg["seed"] = cfg["seed"]
    create_env = get_rlgames_env_creator(task_cfg, cfg["task_name"], cfg["sim_device"], cfg["rl_device"],
                                         cfg["graphics_device_id"], cfg["headless"], multi_gpu=multi_gpu)
    if vec_env_factory is not None:
        if multi_gpu:  # what create_rlgpu_env records for a rank (rlgames_utils.py:71-81 + global env ids)
            task_cfg["rank"] = rank
            task_cfg["env_id_offset"] = rank * int(task_cfg["env"]["numEnvs"])
        vec_env = vec_env_factory(task_cfg, rank)
    else:
        vec_env = RLGPUEnv("rlgpu", task_cfg["env"]["numEnvs"], env_creator=create_env)
    params = cfg["train"]["params"]
    name = params["config"]["name"]
    run_dir = os.path.join("runs", str(name))
    writer = None
    if rank == 0:
        os.makedirs(run_dir, exist_ok=True)
        with open(os.path.join(run_dir, "config.yaml"), "w") as f:
            yaml.safe_dump(cfg, f)
        writer = RLGPUAlgoObserver(run_dir)
    rl_device = vec_env.env.rl_device if hasattr(vec_env, "env") else "cpu"
    agent = A2CAgent(params, vec_env, rl_device, writer=writer, rank=rank, world=world if multi_gpu else 1)
    if params.get("load_checkpoint") and not cfg.get("test"):
        agent.restore(params["load_path"])
    if cfg.get("test"):
        result = play(cfg, vec_env, params)
        if multi_gpu:
            dist.barrier()
            dist.destroy_process_group()
        return result
    result = agent.train()
    if multi_gpu:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    launch()
