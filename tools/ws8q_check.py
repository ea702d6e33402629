#!/usr/bin/env python3
"""ws8q (four lanes per env) against ws8 from identical states, resynchronised every step: largest differences per quantity, and the
step time of both on this box (full stores and BEZ_FLAG_LEAN_STEP).   python tools/ws8q_check.py [variant ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from tests.sim_adapter import SimAdapter
from tests.test_tasks import make_cfg
from bez_isaacgym_amd import abi


def pair(n, **kw):
    os.environ["BEZ_SIM_KERNEL"] = "ws8"
    a = SimAdapter(make_cfg(n, **kw))
    os.environ["BEZ_SIM_KERNEL"] = "ws8q"
    b = SimAdapter(make_cfg(n, **kw))
    os.environ.pop("BEZ_SIM_KERNEL", None)
    return a, b


for variant in (sys.argv[1:] or ["kick", "kick_cleats", "walk"]):
    n = 200
    task = "bez_walk" if variant.startswith("walk") else ("bez_orient" if variant.startswith("orient") else "bez_kick")
    a, b = pair(n, seed=31, task=task, cleats=variant.endswith("_cleats"), box=variant.endswith("_box"))
    rng = np.random.default_rng(8)
    worst = {}
    for t in range(40):
        b.set_root_states(a.root_states); b.set_dof_state(a.dof_state); b.set_contact_forces(a.contact_forces)
        b.set_targets(a.targets); b.set_reset(a.reset_buf); b.set_progress(a.progress_buf); b.set_prev_lin_vel(a.prev_lin_vel)
        if task != "bez_kick": b.set_goal(a.goal)
        act = rng.uniform(-1, 1, (n, 18)).astype(np.float32)
        a.step(act); b.step(act)
        for k in ("root_states", "dof_state", "contact_forces", "obs", "rew", "reset_buf", "progress_buf", "feet", "targets", "prev_lin_vel"):
            x, y = np.asarray(getattr(a, k), np.float64), np.asarray(getattr(b, k), np.float64)
            worst[k] = max(worst.get(k, 0.0), float(np.nanmax(np.abs(x - y))) if x.size else 0.0)
            if np.isnan(y).any() and not np.isnan(x).any(): worst[k] = float("nan")
    print(variant, " ".join("%s %.3g" % kv for kv in worst.items()), flush=True)

# timing: 4096 envs, the bench's random actions, the two kernels interleaved
from bez_isaacgym_amd import build
N, STEPS, ROUNDS = 4096, 1500, 3
L = C.CDLL(build.lib_path())
sims = []
for kern in ("ws8", "ws8q"):
    for lean in (1, 0):
        os.environ["BEZ_SIM_KERNEL"] = kern
        cfg = abi.default_config(N)
        if lean: cfg.flags |= abi.FLAG_LEAN_STEP
        h = C.c_void_p()
        assert L.bez_sim_create(C.byref(cfg), 0, C.byref(h)) == 0
        sims.append(("%s %s" % (kern, "lean" if lean else "full"), h))
os.environ.pop("BEZ_SIM_KERNEL", None)
acts = (torch.rand(64, N * 18, device="cuda") * 2 - 1).contiguous()
res = {k: [] for k, _ in sims}
for r in range(ROUNDS):
    for k, h in sims:
        for t in range(100): L.bez_sim_step(h, C.c_void_p(acts[t % 64].data_ptr()), None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(STEPS): L.bez_sim_step(h, C.c_void_p(acts[t % 64].data_ptr()), None)
        e1.record(); torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) * 1e3 / STEPS)
for k in res: print("%-12s %s  mean %.3f us" % (k, " ".join("%.3f" % x for x in res[k]), float(np.mean(res[k]))), flush=True)

# the other kernel variants: cleats asset (per-env parameter loads compiled in), walk task
for label, kw in (("cleats", dict(cleats=True)), ("walk", dict(task="bez_walk")), ("box", dict(box=True))):
    sims = []
    for kern in ("ws8", "ws8q"):
        os.environ["BEZ_SIM_KERNEL"] = kern
        cfg = make_cfg(N, seed=5, **kw); cfg.flags |= abi.FLAG_LEAN_STEP
        h = C.c_void_p()
        assert L.bez_sim_create(C.byref(cfg), 0, C.byref(h)) == 0
        sims.append(("%s %s" % (kern, label), h))
    os.environ.pop("BEZ_SIM_KERNEL", None)
    res = {k: [] for k, _ in sims}
    for r in range(ROUNDS):
        for k, h in sims:
            for t in range(100): L.bez_sim_step(h, C.c_void_p(acts[t % 64].data_ptr()), None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for t in range(STEPS): L.bez_sim_step(h, C.c_void_p(acts[t % 64].data_ptr()), None)
            e1.record(); torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) * 1e3 / STEPS)
    for k in res: print("%-14s %s  mean %.3f us" % (k, " ".join("%.3f" % x for x in res[k]), float(np.mean(res[k]))), flush=True)
