#!/usr/bin/env python3
"""Step time of the fused control step against the number of workgroups in flight: ws8 (64 envs per workgroup) and ws8q (16 envs per
workgroup) at several num_envs, lean steps, HIP events.   python tools/ws_scale_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from bez_isaacgym_amd import abi, build
L = C.CDLL(build.lib_path())
STEPS = 1000
for kern, per in (("ws8", 64), ("ws8q", 16)):
    for N in (256, 1024, 2048, 4096, 8192, 16384):
        os.environ["BEZ_SIM_KERNEL"] = kern
        cfg = abi.default_config(N); cfg.flags |= abi.FLAG_LEAN_STEP
        h = C.c_void_p()
        assert L.bez_sim_create(C.byref(cfg), 0, C.byref(h)) == 0
        acts = (torch.rand(16, N * 18, device="cuda") * 2 - 1).contiguous()
        ts = []
        for r in range(3):
            for t in range(100): L.bez_sim_step(h, C.c_void_p(acts[t % 16].data_ptr()), None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for t in range(STEPS): L.bez_sim_step(h, C.c_void_p(acts[t % 16].data_ptr()), None)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / STEPS)
        print("%-5s N %6d  workgroups %4d  %.2f us per step  (%.3g env-steps/s)" % (kern, N, (N + per - 1) // per, min(ts), N / min(ts) * 1e6), flush=True)
        L.bez_sim_destroy(h)
