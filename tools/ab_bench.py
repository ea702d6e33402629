#!/usr/bin/env python3
"""A/B timing of the fused control step between library builds on the SAME box (devices differ by several percent, so numbers
from different gpurun calls are not comparable):  python tools/ab_bench.py build_ab/a.so build_ab/b.so ...
Interleaves the libraries over several rounds; prints the mean microseconds per step of each (HIP events)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from bez_isaacgym_amd import abi
N, STEPS, ROUNDS = 4096, 1500, 4
libs = []
for spec in sys.argv[1:]:  # build_ab/x.so or build_ab/x.so:lane (BEZ_SIM_KERNEL for that sim)
    path, _, kern = spec.partition(":")
    os.environ.pop("BEZ_SIM_KERNEL", None)
    if kern: os.environ["BEZ_SIM_KERNEL"] = kern
    lib = C.CDLL(os.path.abspath(path))
    cfg = abi.default_config(N)
    h = C.c_void_p()
    assert lib.bez_sim_create(C.byref(cfg), 0, C.byref(h)) == 0, path
    libs.append((spec, lib, h))
SCALE = float(os.environ.get("AB_ACTION_SCALE", "1.0"))  # 0 = standing (no leg<->leg contacts), 1 = the bench's uniform random actions
acts = ((torch.rand(64, N * 18, device="cuda") * 2 - 1) * SCALE).contiguous()
res = {p: [] for p, _, _ in libs}
for r in range(ROUNDS):
    for path, lib, h in libs:
        for t in range(100):
            lib.bez_sim_step(h, C.c_void_p(acts[t % 64].data_ptr()), None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(STEPS):
            lib.bez_sim_step(h, C.c_void_p(acts[t % 64].data_ptr()), None)
        e1.record(); torch.cuda.synchronize()
        res[path].append(e0.elapsed_time(e1) * 1e3 / STEPS)
for path in res:
    print("%-28s %s  mean %.3f us" % (os.path.basename(path), " ".join("%.3f" % x for x in res[path]), float(np.mean(res[path]))))
