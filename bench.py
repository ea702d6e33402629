#!/usr/bin/env python3
"""bez_kick random-action rollout benchmark (BASELINE.json configs[1]): env-steps/s at num_envs=4096 per GPU.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One process per GPU; every rank owns its own 4096 envs (weak scaling; env ids are global so reset noise does not
depend on N); the rollout has no data-path collective.  A "step" is one fused control step of all local envs
(bez_sim_step: action clamp, PD targets, 2 physics substeps, bookkeeping, reset, obs, reward) on synthetic
U(-1,1) actions that are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = 828  # SURVEY.md 8(d) / BASELINE.md 4: read 336 B + write 492 B per env-step
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
ACTION_RING = 64               # distinct pre-generated action batches cycled through


def pmc_traffic(num_envs):
    """HBM bytes per launch of the dominant kernel from a committed rocprofv3 PMC summary (separate FETCH_SIZE / WRITE_SIZE
    passes, corrected as MI355X_MICROARCH.md prescribes; tools/collect_profiles.sh + tools/summarize_profiles.py).  bench.py
    cannot run under the profiler itself, so this is the profiled figure -- but ONLY of a profile whose recorded source hash
    is the hash of the library that is running now and whose size matches; otherwise None (never a stale number)."""
    import glob
    from bez_isaacgym_amd.build import source_hash
    cur = source_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
            if d.get("source_hash") == cur and int(d["num_envs"]) == int(num_envs):
                return float(d["hbm_bytes_per_launch"]["total"]), os.path.basename(path)
        except Exception:
            continue
    return None, None


def cpu_baseline(num_envs, seconds_target=12.0):
    """The oracle (kind 'port': this build's CPU restatement; the reference's own CPU pipeline is the closed
    PhysX binary and cannot run) on the host cores, bounded sample of the same workload."""
    import numpy as np
    from oracle.bez_oracle import Oracle, build
    from bez_isaacgym_amd import abi
    build()
    cores = min(len(os.sched_getaffinity(0)), 16)  # a 1-GPU box's CPU share is 16 cores
    os.environ["OMP_NUM_THREADS"] = str(cores)  # read by libgomp when the oracle library is first loaded
    orc = Oracle(abi.default_config(num_envs))
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, (8, num_envs, 18)).astype(np.float32)
    orc.step(acts[0])
    t0 = time.perf_counter()
    orc.step(acts[1])
    one = time.perf_counter() - t0
    steps = int(max(6, min(2000, seconds_target / max(one, 1e-6))))
    seg, rates, dt = max(steps // 4, 1), [], 0.0
    for k in range(4):  # four segments: the host cores of a shared box are noisy, report the spread with the mean
        t0 = time.perf_counter()
        for t in range(seg):
            orc.step(acts[t % 8])
        d = time.perf_counter() - t0
        dt += d
        rates.append(num_envs * seg / d)
    return {"value": num_envs * seg * 4 / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "spread": [min(rates), max(rates)],
            "sample": "%d envs x %d control steps in 4 segments, fp64 C oracle, OpenMP over envs (%.1f s)" % (num_envs, seg * 4, dt)}


def ppo_leg(args, rank, local_rank, world, n):
    """bez_kick PPO training (BASELINE.json configs[2]: rl_games default MLP, horizon 32, minibatch 32768, 5 mini-epochs,
    AMP): samples/s = horizon * envs * GPUs / epoch wall time (rollout + update), max over ranks."""
    import torch
    import torch.distributed as dist
    from bez_isaacgym_amd.ppo.a2c_continuous import A2CAgent
    from bez_isaacgym_amd.utils.config import load_config
    from bez_isaacgym_amd.utils.rlgames_utils import RLGPUEnv, get_rlgames_env_creator
    cfg = load_config(["task=bez_kick", "num_envs=%d" % n, "headless=True"])
    task = cfg["task"]
    task["seed"] = 42
    dev = "cuda:%d" % local_rank
    task["rl_device"] = dev
    task["env_id_offset"] = rank * n
    venv = RLGPUEnv("rlgpu", n, env_creator=get_rlgames_env_creator(task, "bez_kick", dev, dev, 0, True))
    params = cfg["train"]["params"]
    params["config"].update(save_frequency=0, save_best_after=10 ** 9)
    params["config"]["minibatch_size"] = min(int(params["config"]["minibatch_size"]), n * int(params["config"]["horizon_length"]))
    agent = A2CAgent(params, venv, dev, rank=rank, world=world)
    agent.obs = agent.env_reset()
    for _ in range(4):  # warm-up: allocations, hipBLASLt heuristics, 2 eager epochs, then HIP-graph capture of rollout + update
        agent.train_epoch()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    play = 0.0
    for _ in range(args.ppo_epochs):
        st = agent.train_epoch()
        play += st["play_time"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt, play], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, play = float(t[0]), float(t[1])
    samples = agent.batch_size * world * args.ppo_epochs
    return {"metric": "PPO samples/s (rollout + update)", "value": samples / dt, "unit": "samples/s", "epochs": args.ppo_epochs,
            "samples_per_epoch": agent.batch_size * world, "epoch_ms": dt / args.ppo_epochs * 1e3,
            "rollout_share": play / dt, "dtype": "fp16 autocast (AMP, as bez_kickPPO.yaml mixed_precision) + f32 sim",
            "minibatch": agent.minibatch_size, "mini_epochs": agent.mini_epochs}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--num-envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ppo-epochs", type=int, default=10, help="extra leg after the timed rollout: PPO epochs timed for the "
                    "'PPO samples/s' half of BASELINE.json's metric (0 = skip)")
    ap.add_argument("--launch", choices=["python", "c"], default="python",
                    help="python: one bez_sim_step call per step from Python; c: bez_sim_step_many (C loop)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from bez_isaacgym_amd import abi
    from bez_isaacgym_amd.sim import BezSim

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    n = args.num_envs
    sim = BezSim(abi.default_config(n, seed=42, env_id_offset=rank * n), local_rank)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1000 + rank)
    actions = (torch.rand(ACTION_RING, n * 18, device=dev, generator=gen) * 2 - 1).contiguous()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(k, t0=0):
        if args.launch == "c":
            done = 0
            while done < k:  # the ring is contiguous: up to ACTION_RING steps per C call
                start = (t0 + done) % ACTION_RING
                m = min(k - done, ACTION_RING - start)
                sim.step_many(actions[start:start + m].reshape(-1), m)
                done += m
        else:
            for t in range(k):
                sim.step(actions[(t0 + t) % ACTION_RING])

    run(args.warmup)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_start = time.perf_counter()
    ev0.record()  # same stream the kernels are launched on (torch's current stream is passed through the C ABI)
    run(args.steps, args.warmup)
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t_start
    dev_ms = ev0.elapsed_time(ev1)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # sanity: the rollout really ran (resets happen with random actions; everything finite)
    obs = sim.tensor(abi.TENSOR_OBS)
    assert torch.isfinite(obs).all()

    ppo = None
    if args.ppo_epochs > 0:
        ppo = ppo_leg(args, rank, local_rank, world, n)

    if rank == 0:
        total_envs = n * world
        kernel_ms = dev_ms / args.steps  # avg device time per fused-step launch over the timed region (HIP events)
        traffic, traffic_src = pmc_traffic(n)
        achieved = ALGO_BYTES_PER_ENV_STEP * n / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "env-steps/s at num_envs=4096 (random-action rollout, bez_kick)",
            "value": total_envs * args.steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "bez_kick num_envs=%d per GPU, random-action rollout only (no PPO), dt=1/60 s x 2 substeps, "
                                   "natural resets included" % n,
                       "envs_per_gpu": n, "total_envs": total_envs, "launch": args.launch, "parallelism": "env-sharded x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "bez::w8::step_kernel_ws8<true,true,false,false> (fused control step, default asset, 8 role waves per 64 envs)", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * n,
                         "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE: instruction stream x1 + data x2, + WRITE_SIZE; source: profiles/%s)" % traffic_src
                                         if traffic is not None else "no committed PMC profile matches this build's source hash",
                         "note": "N=4096 is latency-bound (64 workgroups x 8 role waves on 256 CUs, serial ABA chains): see DESIGN.md"},
        }
        if ppo is not None:
            out["ppo"] = ppo
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(n)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
