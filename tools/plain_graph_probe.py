#!/usr/bin/env python3
"""VERDICT round 3, item 4 / ADVICE (medium): the plain torch PPO path (fused_ops: False) replayed as HIP graphs under AMP stops learning
once agent.save() runs between replays (profiles/r03_plain_graph_bisect.txt).  This probe drives that configuration from a harness and
varies WHAT happens between replays, to find which part of a checkpoint save disturbs the replayed update:

    none            nothing between replays (control)
    save            agent.save(path) every --every epochs (the failing case)
    sync            torch.cuda.synchronize() only
    state           state = agent.get_full_state_weights() only (no serialisation)
    d2h             every tensor of the state copied to the host (.cpu()), nothing pickled
    d2h_pinned      the same through a pinned staging buffer (non_blocking + synchronize)
    save_clone      torch.save of a state whose device tensors were first CLONED on the device (the copies to the host read the clones)
    alloc           a 64 MB device allocation + free (allocator traffic only)

For every variant: epochs until the first spurious non-finite step, number of loss-scale back-offs, final scale, and the device
addresses of the tensors the captured update touches (GradScaler scale / growth tracker / found_inf, Adam step counters, gradients)
before and after the first disturbance, with the allocator's segment list (pool ids) to tell whether anything moved into or out of
the graph's private pool.

    python tools/plain_graph_probe.py --variants none save sync state d2h --epochs 400 > profiles/r04_plain_graph_probe.txt
"""
import argparse
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_agent(num_envs, seed):
    import torch
    from bez_isaacgym_amd.ppo.a2c_continuous import A2CAgent
    from bez_isaacgym_amd.utils.config import load_config
    from bez_isaacgym_amd.utils.rlgames_utils import RLGPUEnv, get_rlgames_env_creator
    from bez_isaacgym_amd.utils.utils import set_seed
    import warnings
    warnings.simplefilter("ignore")
    cfg = load_config(["task=bez_kick", "num_envs=%d" % num_envs, "headless=True", "seed=%d" % seed])
    set_seed(seed)
    cfg["task"]["seed"] = seed
    venv = RLGPUEnv("rlgpu", num_envs, env_creator=get_rlgames_env_creator(cfg["task"], "bez_kick", "cuda:0", "cuda:0", 0, True))
    params = cfg["train"]["params"]
    params["config"].update(save_frequency=0, save_best_after=10 ** 9, fused_ops=os.environ.get("BEZ_PROBE_FUSED") == "1", hip_graphs=True)
    if os.environ.get("BEZ_PROBE_TORCH_CLIP") == "1":   # put torch's own clip_grad_norm_ (torch.stack inside) back: the failing configuration
        import bez_isaacgym_amd.ppo.a2c_continuous as M
        M.clip_grad_norm_capturable = lambda ps, mx: torch.nn.utils.clip_grad_norm_(list(ps), mx)
    params["config"]["minibatch_size"] = min(int(params["config"]["minibatch_size"]), num_envs * int(params["config"]["horizon_length"]))
    return A2CAgent(params, venv, "cuda:0")


def tensors_of(state, out=None, prefix=""):
    import torch
    out = [] if out is None else out
    if torch.is_tensor(state):
        out.append((prefix, state))
    elif isinstance(state, dict):
        for k, v in state.items():
            tensors_of(v, out, "%s/%s" % (prefix, k))
    elif isinstance(state, (list, tuple)):
        for i, v in enumerate(state):
            tensors_of(v, out, "%s/%d" % (prefix, i))
    return out


def addresses(agent):
    sc = agent.scaler
    d = {"scale": sc._scale.data_ptr(), "growth_tracker": sc._growth_tracker.data_ptr()}
    for i, st in enumerate(sc._per_optimizer_states.values()):
        for dev, t in st.get("found_inf_per_device", {}).items():
            d["found_inf[%d]" % i] = t.data_ptr()
    ps = list(agent.model.parameters())
    d["grad[0]"] = ps[0].grad.data_ptr() if ps[0].grad is not None else 0
    d["grad[-1]"] = ps[-1].grad.data_ptr() if ps[-1].grad is not None else 0
    st0 = agent.optimizer.state.get(ps[0], {})
    if st0:
        d["adam_step[0]"] = st0["step"].data_ptr()
        d["exp_avg[0]"] = st0["exp_avg"].data_ptr()
    return d


def pool_summary():
    import torch
    snap = torch.cuda.memory_snapshot()
    pools = {}
    for seg in snap:
        key = tuple(seg.get("segment_pool_id", (0, 0)))
        p = pools.setdefault(key, dict(segments=0, bytes=0, active=0, inactive=0))
        p["segments"] += 1; p["bytes"] += seg["total_size"]
        for b in seg["blocks"]:
            p["active" if b["state"].startswith("active") else "inactive"] += 1
    return pools


def disturb(agent, variant, path):
    import torch
    if variant == "none":
        return
    if variant == "sync":
        torch.cuda.synchronize(); return
    if variant == "alloc":
        t = torch.empty(16 << 20, device="cuda"); del t; return
    if variant == "save":
        agent.save(path); return
    if variant == "scaler_sd":
        agent.scaler.state_dict(); return
    if variant == "opt_sd":
        agent.optimizer.state_dict(); return
    if variant == "model_sd":
        agent.model.state_dict(); agent.running_mean_std.state_dict(); agent.value_mean_std.state_dict(); return
    if variant == "item":
        agent.scaler._scale.item(); return
    if variant == "pinned_churn":   # the suspected mechanism, provoked: recycle the pinned-host allocator's small blocks with junk
        junk = [torch.empty(sz, dtype=torch.uint8, pin_memory=True).fill_(255) for sz in (512, 1024, 2048, 4096, 8192, 16384) for _ in range(32)]
        del junk; return
    if variant == "alloc_fill":     # eager device allocations of assorted sizes, written with NaN, freed again: nothing of the agent is touched
        junk = [torch.full((n,), float("nan"), device="cuda") for n in (1, 16, 128, 1024, 4096, 21600, 80000, 124237, 1 << 20) for _ in range(3)]
        del junk; return
    if variant == "param_double":
        junk = [p.detach().double() for p in agent.model.parameters()]; del junk; return
    if variant == "param_sum":
        junk = [p.detach().sum() for p in agent.model.parameters()]; del junk; return
    if variant in ("small_sum", "big_sum", "small_max", "big_copy"):   # reductions / copies over tensors that have nothing to do with the agent
        x = getattr(agent, "_probe_x", None)
        if x is None:
            x = agent._probe_x = (torch.ones(100, device="cuda"), torch.ones(1 << 20, device="cuda"))
        junk = x[0].sum() if variant == "small_sum" else x[1].sum() if variant == "big_sum" else x[0].max() if variant == "small_max" else x[1].clone()
        del junk; return
    if variant.startswith("sync_"):   # the same disturbance, but only after everything enqueued so far has finished
        torch.cuda.synchronize()
        return disturb(agent, variant[5:], path)
    if variant.startswith("rep_"):    # N launches of one small kernel on a tensor that has nothing to do with the agent
        _, op, n = variant.split("_")
        x = getattr(agent, "_probe_y", None)
        if x is None:
            x = agent._probe_y = torch.ones(100, device="cuda")
        for _ in range(int(n)):
            junk = x.sum() if op == "sum" else x.clone() if op == "copy" else x.mul(2.0) if op == "mul" else torch.cumsum(x, 0)
        del junk; return
    if variant.startswith("psum_"):   # which parameter, which reduction
        ps = list(agent.model.parameters())
        what = variant[5:]
        if what == "clone":
            junk = [p.detach().clone().sum() for p in ps]
        elif what == "max":
            junk = [p.detach().max() for p in ps]
        elif what == "mean":
            junk = [p.detach().mean() for p in ps]
        elif what == "seq":      # one at a time, nothing kept alive
            for p in ps:
                junk = p.detach().sum()
        elif what == "w":        # weight matrices only
            junk = [p.detach().sum() for p in ps if p.dim() == 2]
        elif what == "b":        # vectors only
            junk = [p.detach().sum() for p in ps if p.dim() == 1]
        else:
            junk = [ps[int(what)].detach().sum()]
        del junk; return
    if variant == "empty_d2h":
        torch.zeros(4, device="cuda").cpu(); return
    state = agent.get_full_state_weights()
    if variant == "state":
        return
    ts = tensors_of(state)
    if variant == "d2h":
        _ = [t.cpu() for _, t in ts if t.is_cuda]; return
    if variant == "d2h_pinned":
        outs = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True).copy_(t, non_blocking=True) for _, t in ts if t.is_cuda]
        torch.cuda.synchronize(); del outs; return
    if variant == "save_clone":
        def clone(o):
            if torch.is_tensor(o):
                return o.clone()
            if isinstance(o, dict):
                return {k: clone(v) for k, v in o.items()}
            if isinstance(o, (list, tuple)):
                return type(o)(clone(v) for v in o)
            return o
        torch.save(clone(state), io.BytesIO()); return
    raise ValueError(variant)


def run(variant, epochs, every, num_envs, seed, checks=None):
    import torch
    agent = make_agent(num_envs, seed)
    agent.obs = agent.env_reset()
    path = "/tmp/plain_graph_probe_%s.pth" % variant
    scales, events, first = [], 0, None
    before = after = pools_b = pools_a = None
    t0 = time.time()
    for ep in range(1, epochs + 1):
        agent.train_epoch()
        if ep % every == 0 and agent._g_update is not None:
            if before is None:
                torch.cuda.synchronize()
                before, pools_b = addresses(agent), pool_summary()
            disturb(agent, variant, path)
            if after is None:
                torch.cuda.synchronize()
                after, pools_a = addresses(agent), pool_summary()
        if checks is not None and (ep % every in (0, 1, 2) or ep % every == every - 1) and ep <= 3 * every + 2:
            checks.append((ep, float(sum(p.detach().double().sum() for p in agent.model.parameters()))))
        s = float(agent.scaler.get_scale())
        if scales and s < scales[-1]:
            events += 1
            first = first or ep
        scales.append(s)
    rew = agent.game_rewards[0] if agent.game_rewards else float("nan")
    finite = all(bool(torch.isfinite(p).all()) for p in agent.model.parameters())
    moved = {k: (hex(before[k]), hex(after[k])) for k in before if before[k] != after.get(k)} if before else {}
    print("%-11s epochs %d  first back-off at epoch %s  back-offs %d  final scale %g  weights finite %s  mean reward %.2f  (%.0f s)" % (
        variant, epochs, first, events, scales[-1], finite, rew, time.time() - t0), flush=True)
    if checks:
        print("            parameter checksums: " + "  ".join("%d:%.10f" % c for c in checks))
    if before and checks is None:
        print("            addresses that moved across the first disturbance: %s" % (moved or "none"))
        print("            allocator pools before: %s" % {str(k): v for k, v in pools_b.items()})
        print("            allocator pools after : %s" % {str(k): v for k, v in pools_a.items()})
    del agent
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", nargs="+", default=["none", "save", "sync", "state", "d2h"])
    ap.add_argument("--epochs", type=int, default=400)
    ap.add_argument("--every", type=int, default=20)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--checksums", action="store_true", help="parameter checksums around the first disturbances instead of the address report")
    a = ap.parse_args()
    for v in a.variants:
        run(v, a.epochs, a.every, a.envs, a.seed, [] if a.checksums else None)


if __name__ == "__main__":
    main()
