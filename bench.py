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


def cpu_baseline(num_envs, seconds_target=9.0, seconds_single=3.0, seconds_small=2.0):
    """The oracle (kind 'port': this build's CPU restatement; the reference's own CPU pipeline is the closed
    PhysX binary and cannot run) on the host cores, bounded sample of the same workload: all cores of the box's
    share, then one thread (BASELINE.md 3 asks for both), with mean / p50 / p99 step times."""
    import ctypes
    import numpy as np
    from oracle.bez_oracle import Oracle, build
    from bez_isaacgym_amd import abi
    build()
    cores = min(len(os.sched_getaffinity(0)), 16)  # a 1-GPU box's CPU share is 16 cores
    os.environ["OMP_NUM_THREADS"] = str(cores)  # read by libgomp when the oracle library is first loaded
    orc = Oracle(abi.default_config(num_envs))
    gomp = ctypes.CDLL("libgomp.so.1")
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, (8, num_envs, 18)).astype(np.float32)

    def sample(threads, budget, orc=orc, acts=acts, num_envs=num_envs):
        gomp.omp_set_num_threads(threads)
        orc.step(acts[0])
        t0 = time.perf_counter()
        orc.step(acts[1])
        one = time.perf_counter() - t0
        steps = int(max(8, min(2000, budget / max(one, 1e-6))))
        seg, rates, times = max(steps // 4, 2), [], []
        for k in range(4):  # four segments: the host cores of a shared box are noisy, report the spread with the mean
            t0 = time.perf_counter()
            for t in range(seg):
                t1 = time.perf_counter()
                orc.step(acts[t % 8])
                times.append(time.perf_counter() - t1)
            rates.append(num_envs * seg / (time.perf_counter() - t0))
        times = np.array(times)
        return {"value": num_envs * times.size / float(times.sum()), "spread": [min(rates), max(rates)], "steps": int(times.size), "seconds": float(times.sum()),
                "ms_per_step": {"mean": float(times.mean() * 1e3), "p50": float(np.percentile(times, 50) * 1e3), "p99": float(np.percentile(times, 99) * 1e3)}}
    multi = sample(cores, seconds_target)
    single = sample(1, seconds_single)
    # BASELINE.md 3 asks for N = 64 as well (BASELINE.json configs[0], the reference's own CPU-runnable size)
    small_args = dict(orc=Oracle(abi.default_config(64)), acts=rng.uniform(-1, 1, (8, 64, 18)).astype(np.float32), num_envs=64)
    small = sample(cores, seconds_small, **small_args)
    small1 = sample(1, seconds_small, **small_args)
    return {"value": multi["value"], "unit": "env-steps/s", "cores": cores, "kind": "port", "spread": multi["spread"],
            "ms_per_step": multi["ms_per_step"],
            "single_thread": {"value": single["value"], "cores": 1, "ms_per_step": single["ms_per_step"],
                              "sample": "%d control steps (%.1f s)" % (single["steps"], single["seconds"])},
            "num_envs_64": {"value": small["value"], "cores": cores, "ms_per_step": small["ms_per_step"],
                            "single_thread": {"value": small1["value"], "ms_per_step": small1["ms_per_step"]},
                            "sample": "64 envs x %d / %d control steps (BASELINE.json configs[0])" % (small["steps"], small1["steps"])},
            "sample": "%d envs x %d control steps in 4 segments, fp64 C oracle, OpenMP over envs (%.1f s)" % (num_envs, multi["steps"], multi["seconds"])}


def launch_ranks(n, argv, env=None, python=sys.executable, timeout=1500.0):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one per GPU: RANK = LOCAL_RANK = i,
    WORLD_SIZE = N, rendezvous on 127.0.0.1) BEFORE anything in this process touches a GPU, let rank 0's stdout through
    (its ONE JSON line), and return the worst exit code.  The reference picks the device the same way, from the rank
    (utils/rlgames_utils.py:71-81).  torch.distributed.run does exactly this when the driver uses it; then WORLD_SIZE is
    already set and this function is not reached."""
    import socket
    import subprocess
    # the port is found by binding port 0; SO_REUSEADDR + keeping the probe socket open until the children exist narrows the
    # window in which another process could take it (the ranks' TCPStore binds with SO_REUSEADDR as well)
    probe = socket.socket()
    probe.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    probe.bind(("127.0.0.1", 0))
    port = probe.getsockname()[1]
    procs = []
    try:
        for r in range(n):
            e = dict(os.environ if env is None else env)
            e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([python, os.path.abspath(__file__)] + list(argv), env=e,
                                          stdout=None if r == 0 else subprocess.DEVNULL))
    finally:
        probe.close()
    return wait_ranks(procs, timeout)


def wait_ranks(procs, timeout=1500.0, grace=10.0, poll=0.2):
    """Polls ALL children: the first non-zero exit (import error, OOM, HIP error in one rank) terminates the siblings -- which
    would otherwise sit in the rendezvous or a barrier for ever -- after `grace` seconds kills them, and its code is returned;
    so does the overall `timeout` (code 124).  0 only if every rank exited 0."""
    deadline = time.monotonic() + timeout
    rc = 0
    live = list(procs)
    while live and rc == 0:
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0:
                rc = abs(r) or 1
                break
        else:
            if time.monotonic() > deadline:
                rc = 124
            elif live:
                time.sleep(poll)
    if live:  # a rank failed or the time ran out: take the others down
        for p in live:
            p.terminate()
        t_end = time.monotonic() + grace
        for p in live:
            try:
                p.wait(max(0.0, t_end - time.monotonic()))
            except Exception:
                p.kill()
                p.wait()
    return rc


MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA peak (the sparsity figure is twice that and never used)
VALU_PEAK_GINST_S = 256 * 2 * 2.4  # wave64 vector instructions per second at peak: 256 CUs x 128 fp32 lanes = 2 per CU and clock, 2.4 GHz (= 157 TFLOP/s of FMAs)


def profile_tag_for_this_build(num_envs):
    """Tag ("r05") of the committed profile set whose PMC summary records THIS build's source hash and size, or None."""
    import glob
    from bez_isaacgym_amd.build import source_hash
    cur = source_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
            if d.get("source_hash") == cur and int(d["num_envs"]) == int(num_envs):
                return os.path.basename(path)[:-len("_pmc_traffic.json")], d
        except Exception:
            continue
    return None, None


def ppo_roofline(agent, samples_per_s, num_envs):
    """MFMA roofline of the PPO half of the metric.  Algorithmic flops per sample = 2 x MACs of the MLP x (1 rollout forward +
    mini_epochs x (forward + input-gradient/weight-gradient backward = 3 passes)) -- SURVEY.md 8(d): 2 x 123 500 x 16 = 3.95 Mflop.
    Per-kernel rows come from the committed rocprofv3 kernel stats of the SAME build (source hash recorded beside them), else None."""
    import csv
    net = agent.model.a2c_network
    macs = sum(int(w.shape[0]) * int(w.shape[1]) for w in (p for n_, p in net.named_parameters() if p.dim() == 2))
    bwd_macs = macs - int(next(p for n_, p in net.named_parameters() if p.dim() == 2).numel())   # no input gradient for the first layer
    passes = 1 + 3 * agent.mini_epochs
    fps = 2.0 * macs * passes
    achieved = fps * samples_per_s / 1e12
    out = {"bound": "mfma", "flops_per_sample": fps, "macs_per_forward": macs, "passes_per_sample": passes, "achieved": achieved,
           "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS, "kernels": None}
    tag, _ = profile_tag_for_this_build(num_envs)
    path = os.path.join(ROOT, "profiles", "%s_ppo_kernel_stats.csv" % tag) if tag else None
    if path and os.path.exists(path):
        mb = float(agent.minibatch_size)
        # unique operand bytes per row of a launch (what has to cross HBM / Infinity Cache at least once; weights are < 0.3 MB and stay in L2):
        # hidden widths h, input width d_in, action width A.  fp16 activations, fp32 observations / loss operands / outputs.
        mats = [p for n_, p in net.named_parameters() if p.dim() == 2]
        nh = len(mats) - 2                                  # actor_mlp layers, then the mu and value heads
        d_in = int(mats[0].shape[1]); hid = [int(mats[i].shape[0]) for i in range(nh)]; A = int(mats[nh].shape[0])
        sh = float(sum(hid))
        row_bytes = {
            "policy_forward_kernel<2": 4.0 * d_in + 2.0 * (d_in + sh) + 4.0 * (A + 1),                  # obs in; x0 + every ELU output kept for the backward pass; mu, value out
            "policy_backward_kernel": 2.0 * sh + 4.0 * 4.0 * A + 2.0 * (sh + A + 1) + 4.0 * 4,           # ELU outputs in; old mu / sigma / action / neglogp blocks; gz of every layer + head gradients out
            "wgrad_kernel": 2.0 * (sh + A + 1) + 2.0 * (d_in + sh),                                      # dY of every layer + X of every layer, each read once
            "policy_forward_kernel<1": 4.0 * d_in + 4.0 * (3 * A + 3),                                   # obs in; action, mu, sigma, value, neglogp rows out
        }
        flops = {"policy_forward_kernel<2": ("training forward", 2.0 * macs * mb, mb), "policy_backward_kernel": ("loss + input-gradient chain", 2.0 * bwd_macs * mb, mb),
                 "wgrad_kernel": ("weight gradients (split-K)", 2.0 * macs * mb, mb), "policy_forward_kernel<1": ("rollout forward", 2.0 * macs * float(agent.num_actors), float(agent.num_actors))}
        ridge = MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)      # flop per byte at which the two roofs meet (312)
        rows = []
        for r in csv.DictReader(open(path)):
            for key, (what, fl, nrow) in flops.items():
                if key in r["Name"]:
                    us = float(r["AverageNs"]) * 1e-3
                    by = row_bytes[key] * nrow
                    inten = fl / by
                    gbs = by / (us * 1e-6) / 1e9
                    rows.append({"kernel": what, "avg_us": us, "gflop_per_launch": fl / 1e9, "tflops": fl / (us * 1e-6) / 1e12,
                                 "frac": fl / (us * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS, "calls": int(r["Calls"]),
                                 "unique_mbytes_per_launch": by / 1e6, "flop_per_byte": inten, "bound": "hbm" if inten < ridge else "mfma",
                                 "hbm_gbs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS})
        out["kernels"] = rows
        out["ridge_flop_per_byte"] = ridge
        out["bound_note"] = ("every PPO kernel sits far below the %.0f flop/B ridge (84 - 140 flop/B): the roof that bounds it is HBM, hbm_frac is the honest figure; "
                             "the MFMA fractions are kept beside it" % ridge)
        out["kernels_source"] = "profiles/%s_ppo_kernel_stats.csv (rocprofv3 --kernel-trace --stats of this build: source hash matches profiles/%s_pmc_traffic.json)" % (tag, tag)
    return out


def ppo_leg(args, rank, local_rank, world, n):
    """bez_kick PPO training (BASELINE.json configs[2]: rl_games default MLP, horizon 32, minibatch 32768, 5 mini-epochs,
    AMP): samples/s = horizon * envs * GPUs / epoch wall time (rollout + update), max over ranks."""
    import torch
    import torch.distributed as dist
    from bez_isaacgym_amd.ppo.a2c_continuous import A2CAgent
    from bez_isaacgym_amd.utils.config import load_config
    from bez_isaacgym_amd.utils.rlgames_utils import RLGPUEnv, get_rlgames_env_creator
    cfg = load_config(["task=bez_kick", "num_envs=%d" % n, "headless=True"] + (["task.task.randomize=True"] if args.randomize else []))
    task = cfg["task"]
    task["seed"] = 42
    dev = "cuda:%d" % local_rank
    task["rl_device"] = dev
    task["env_id_offset"] = rank * n
    venv = RLGPUEnv("rlgpu", n, env_creator=get_rlgames_env_creator(task, "bez_kick", dev, dev, 0, True))
    params = cfg["train"]["params"]
    params["config"].update(save_frequency=0, save_best_after=10 ** 9)
    for k in ("dr_prelaunch", "fold_action_noise", "fold_rollout_post", "fused_dataset_prep", "fused_loss_backward", "dp_capture_collectives", "dp_grid_norm", "dp_eager_update", "pipeline_epochs"):   # A/B switches of the rollout loop (default: all on)
        if os.environ.get("BEZ_PPO_" + k.upper()) is not None:
            params["config"][k] = os.environ["BEZ_PPO_" + k.upper()] == "1"
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
    pending = None   # epochs pipelined as A2CAgent.train() runs them: epoch k's report is read while epoch k + 1 is queued
    for _ in range(args.ppo_epochs):
        ticket = agent.train_epoch_launch() if agent.cfg.get("pipeline_epochs", True) else None
        if ticket is None:
            play += agent.train_epoch()["play_time"]
            continue
        if pending is not None:
            play += agent.train_epoch_finish(pending)["play_time"]
        pending = ticket
    if pending is not None:
        play += agent.train_epoch_finish(pending)["play_time"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt, play], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, play = float(t[0]), float(t[1])
    samples = agent.batch_size * world * args.ppo_epochs
    out = {"metric": "PPO samples/s (rollout + update)", "value": samples / dt, "unit": "samples/s", "epochs": args.ppo_epochs,
           "samples_per_epoch": agent.batch_size * world, "epoch_ms": dt / args.ppo_epochs * 1e3,
           "rollout_share": play / dt, "dtype": "fp16 autocast (AMP, as bez_kickPPO.yaml mixed_precision) + f32 sim",
           "randomize": bool(args.randomize), "hip_graphs": bool(agent.use_graphs), "data_parallel_path": bool(agent._segmented),
           "minibatch": agent.minibatch_size, "mini_epochs": agent.mini_epochs}
    if rank == 0:
        out["roofline"] = ppo_roofline(agent, samples / dt, n)
    agent.release_env()
    return out


def dp_path_leg(args, local_rank, n):
    """The data-parallel PPO code path on ONE GPU: a 1-rank RCCL process group + BEZ_PPO_FORCE_DIST=1, so the
    epoch runs as every rank of an N-GPU job runs it -- rollout graph, dataset preparation around its two all-reduces, per optimiser step a
    forward/backward graph, ONE real RCCL all-reduce of the flat gradient, the optimiser graph -- minus the wire time.  What it costs a rank
    over the single-GPU path is `dp_path_epoch_ms - epoch_ms`."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
    os.environ["BEZ_PPO_FORCE_DIST"] = "1"
    import torch
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
    try:
        r = ppo_leg(args, 0, local_rank, 1, n)
    finally:
        dist.destroy_process_group()
        os.environ.pop("BEZ_PPO_FORCE_DIST", None)
    return {"dp_path_epoch_ms": r["epoch_ms"], "dp_path_samples_per_s": r["value"], "dp_path_rollout_share": r["rollout_share"],
            "dp_path_what": "1-rank RCCL group, BEZ_PPO_FORCE_DIST=1: segmented graphs + %d real all-reduce calls per epoch, no wire time" % (r["mini_epochs"] * (131072 // r["minibatch"] if r["minibatch"] else 0) + 2)}


def dp_path_in_child(args, n, limit_s=240):
    """dp_path_leg in a child process with a time limit: the extra leg brings up a process group, and neither an RCCL failure nor a hang there
    may take the headline line down with it.  Returns the leg's dict, or {"dp_path_error": ...}."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--dp-path-child", "--gpus", "1", "--num-envs", str(n), "--ppo-epochs", str(args.ppo_epochs)] + \
          (["--randomize"] if args.randomize else [])
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:   # a port that is free right now (a fixed number may be taken on a shared box)
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK=os.environ.get("LOCAL_RANK", "0"), WORLD_SIZE="1")
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=limit_s, env=env)
        lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
        if r.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"dp_path_error": "child rc %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:])}
    except subprocess.TimeoutExpired:
        return {"dp_path_error": "child exceeded %d s" % limit_s}
    except Exception as e:   # noqa: BLE001 -- an optional leg
        return {"dp_path_error": repr(e)[:300]}


def stub_main(args, rank, world):
    """The multi-rank skeleton of main() on the CPU (gloo), the env step replaced by a no-op: rendezvous, barrier-bracketed timed
    region, MAX over ranks, one JSON line from rank 0.  Exercised by tests/test_bench_launcher.py."""
    import torch
    import torch.distributed as dist
    if os.environ.get("BEZ_BENCH_STUB_FAIL_RANK") == str(rank):
        sys.exit(7)  # launcher self-test: one rank dies before the rendezvous; its siblings must not be left waiting
    if world > 1:
        dist.init_process_group("gloo")
        assert dist.get_world_size() == args.gpus
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    elapsed = time.perf_counter() - t0 + 1e-9
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ranks = torch.zeros(world, dtype=torch.int64)
        ranks[rank] = 1
        dist.all_reduce(ranks)
        assert int(ranks.sum()) == world
    if rank == 0:
        out = {"metric": "stub", "value": args.num_envs * world * args.steps / elapsed, "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "scaling": "weak", "config": {"parallelism": "env-sharded x%d" % world}}
        if world > 1:
            out["collective"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rank_sum_check": True}
            out["cpu_baseline"] = "N=1 only"
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--num-envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ppo-epochs", type=int, default=30, help="extra leg after the timed rollout: PPO epochs timed for the "
                    "'PPO samples/s' half of BASELINE.json's metric (0 = skip)")
    ap.add_argument("--launch", choices=["python", "c"], default="python",
                    help="python: one bez_sim_step call per step from Python; c: bez_sim_step_many (C loop)")
    ap.add_argument("--keep-aux", action="store_true", help="keep NET_CONTACT_FORCE / FEET / PREV_LIN_VEL current every step (no BEZ_FLAG_LEAN_STEP)")
    ap.add_argument("--no-full-store", action="store_true", help="skip the side measurement of the full-store (non-lean) step: profiler passes "
                    "that average a counter over every dispatch of the step kernel want the headline configuration only")
    ap.add_argument("--dp-path", dest="dp_path", action="store_true", default=True,
                    help="N = 1 (default on): time the PPO leg a second time on the data-parallel code path (1-rank RCCL group, BEZ_PPO_FORCE_DIST=1) and "
                         "report ppo.dp_path_epoch_ms beside ppo.epoch_ms.  Runs in a child process under a time limit: a failure or a hang of that "
                         "extra leg is reported in the line (ppo.dp_path_error), it never costs the headline")
    ap.add_argument("--no-dp-path", dest="dp_path", action="store_false")
    ap.add_argument("--dp-path-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--randomize", action="store_true", help="PPO leg with task.randomize=True (BASELINE.json configs[4]: domain-randomised "
                    "friction / gains / limits / gravity + observation and action noise, redrawn on the device at reset time)")
    ap.add_argument("--stub-cpu", action="store_true", help="launcher self-test: gloo on the CPU, the env step replaced by a no-op "
                    "(tests/test_bench_launcher.py); prints the same JSON skeleton with metric 'stub'")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))  # nothing above this line has touched a GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (python bench.py --gpus N does)" % (args.gpus, world))
    if args.stub_cpu:
        return stub_main(args, rank, world)
    if args.dp_path_child:   # the extra leg of --dp-path, in its own process: one JSON dict on stdout
        import bez_isaacgym_amd  # noqa: F401 -- BEFORE the process group initialises HIP: the package switches the runtime's graph packet capture off
        print(json.dumps(dp_path_leg(args, local_rank, args.num_envs)), flush=True)
        return 0

    import torch
    import torch.distributed as dist
    from bez_isaacgym_amd import abi
    from bez_isaacgym_amd.sim import BezSim

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    n = args.num_envs
    cfg = abi.default_config(n, seed=42, env_id_offset=rank * n)
    if not args.keep_aux:
        # the rollout reads obs / reward / reset only: BEZ_FLAG_LEAN_STEP skips the Isaac-visible extras nothing on this path reads
        # (net contact force rows, feet, prev_lin_vel: 308 of the 912 B an env-step writes; the obs' feet flags are unaffected)
        cfg.flags |= abi.FLAG_LEAN_STEP
    sim = BezSim(cfg, local_rank)
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
    # step-time distribution (mean / p50 / p99), measured AFTER the timed region so that the per-step events do not
    # sit inside it: 400 steps, one HIP event pair each, read back at the end
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(401)]
    evs[0].record()
    for t in range(400):
        sim.step(actions[t % ACTION_RING])
        evs[t + 1].record()
    torch.cuda.synchronize()
    per = sorted(evs[t].elapsed_time(evs[t + 1]) for t in range(400))
    step_ms = {"mean": sum(per) / len(per), "p50": per[200], "p99": per[396], "what": "400 extra steps outside the timed region, HIP event pair per step"}

    # the C ABI's default keeps every Isaac-visible tensor current (no BEZ_FLAG_LEAN_STEP): the same build's full-store step,
    # timed beside the headline so that the difference is on record
    other = None
    full_store_ms = lean_event_ms = None
    if not (args.keep_aux or args.no_full_store):
        ocfg = abi.default_config(n, seed=42, env_id_offset=rank * n)
        ocfg.flags = int(cfg.flags) & ~abi.FLAG_LEAN_STEP      # the headline sim's configuration, minus the lean flag
        other = BezSim(ocfg, local_rank)
        for t in range(50):
            other.step(actions[t % ACTION_RING])
        # both sims timed the SAME way in the same region (round-4 advisor finding: an event-timed figure beside a wall-clock one says
        # nothing): event pairs over 100-step blocks, lean / full interleaved, four blocks each
        acc = {"lean": 0.0, "full": 0.0}
        for blk in range(4):
            for name, s_ in (("lean", sim), ("full", other)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for t in range(100):
                    s_.step(actions[(blk * 100 + t) % ACTION_RING])
                e1.record()
                torch.cuda.synchronize()
                acc[name] += e0.elapsed_time(e1)
        lean_event_ms, full_store_ms = acc["lean"] / 400, acc["full"] / 400
        other.close()

    collective = None
    if world > 1:
        # evidence that the ranks really form one RCCL job: a real all-reduce whose result every rank can check
        chk = torch.tensor([float(rank + 1), 1.0], device=dev, dtype=torch.float64)
        dist.all_reduce(chk)
        ok = int(chk[0].item()) == world * (world + 1) // 2 and int(chk[1].item()) == world
        collective = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rank_sum_check": bool(ok),
                      "rank_sum": chk[0].item(), "data_path_collectives_per_env_step": 0,
                      "ppo_collectives_per_epoch": "optimiser steps + 2 (DESIGN.md 7)"}
        assert ok, chk

    ppo = None
    if args.ppo_epochs > 0:
        ppo = ppo_leg(args, rank, local_rank, world, n)
        if args.dp_path and world == 1:
            ppo.update(dp_path_in_child(args, n))
            if "dp_path_epoch_ms" in ppo:
                ppo["dp_path_overhead"] = ppo["dp_path_epoch_ms"] / ppo["epoch_ms"] - 1.0

    if rank == 0:
        total_envs = n * world
        kernel_ms = dev_ms / args.steps  # avg device time per fused-step launch over the timed region (HIP events)
        traffic, traffic_src = pmc_traffic(n)
        achieved = ALGO_BYTES_PER_ENV_STEP * n / (kernel_ms * 1e-3) / 1e9
        # "also report FP32 VALU fraction" -- wave-level vector instructions per launch (SQ_INSTS_VALU of the hash-matched
        # PMC profile) over this run's kernel time, against the chip's vector issue peak
        valu = None
        _tag, _pmc = profile_tag_for_this_build(n)
        if _pmc is not None and "sq_counters_per_launch" in _pmc:
            vi = float(_pmc["sq_counters_per_launch"]["SQ_INSTS_VALU"])
            ginst = vi / (kernel_ms * 1e-3) / 1e9
            valu = {"insts_per_launch": vi, "achieved": ginst, "peak": VALU_PEAK_GINST_S, "unit": "G wave-instructions/s", "frac": ginst / VALU_PEAK_GINST_S,
                    "waves": _pmc["sq_counters_per_launch"].get("SQ_WAVES"), "wait_share": float(_pmc["sq_counters_per_launch"]["SQ_WAIT_ANY"]) / float(_pmc["sq_counters_per_launch"]["SQ_WAVE_CYCLES"]),
                    "source": "profiles/%s_pmc_traffic.json" % _tag}
        out = {
            "metric": "env-steps/s at num_envs=4096 (random-action rollout, bez_kick)",
            "value": total_envs * args.steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "step_ms": step_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "bez_kick num_envs=%d per GPU, random-action rollout only (no PPO), dt=1/60 s x 2 substeps, "
                                   "natural resets included" % n,
                       "envs_per_gpu": n, "total_envs": total_envs, "launch": args.launch, "parallelism": "env-sharded x%d" % world,
                       "lean_step": not args.keep_aux, "full_store_step_ms": full_store_ms, "lean_event_ms": lean_event_ms,
                       "lean_vs_full_what": "HIP event pairs over 4 x 100-step blocks each, interleaved, after the timed region (device time; the headline ms_per_step is wall clock)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "valu_frac": None if valu is None else valu["frac"], "valu": valu,
                         "kernel": {"ws8": "bez::w8::step_kernel_ws8<true,true,false,false> (fused control step, default asset, 8 role waves per 64 envs; BEZ_SIM_KERNEL=ws8)",
                                    "lane": "bez::step_kernel<true,true,true,false,false> (one env per lane; BEZ_SIM_KERNEL=lane)"}.get(os.environ.get("BEZ_SIM_KERNEL", ""),
                                    "bez::w8q::step_kernel_ws8<true,true,false,false> (fused control step, default asset, 8 role waves per 16 envs, four lanes per env)"),
                         "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * n,
                         "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE: instruction stream x1 + data x2, + WRITE_SIZE; source: profiles/%s)" % traffic_src
                                         if traffic is not None else "no committed PMC profile matches this build's source hash",
                         "note": "N=4096 is latency-bound (256 workgroups x 8 role waves, serial ABA chains; the instruction stream of a leg role sets the time): see DESIGN.md"},
        }
        if ppo is not None:
            out["ppo"] = ppo
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(n)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = "N=1 only"
        if collective is not None:
            out["collective"] = collective
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
