#!/usr/bin/env python3
"""PMC attribution probe: the fused step at several sizes (1 / 8 / 64 / 256 workgroups) so that the per-launch fixed part of
FETCH_SIZE (instruction fetch of a ~300 KB kernel into each XCD's L2, kernel arguments) separates from the per-env part.
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT -- python3 tools/pmc_probe.py"""
import os, sys
os.environ.setdefault("BEZ_SIM_KERNEL", "ws8q")   # one kernel at every size (unset, the library switches to the one-lane form above 16 x CUs envs)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bez_isaacgym_amd import abi
from bez_isaacgym_amd.sim import BezSim
for n in (64, 512, 4096, 16384):
    cfg = abi.default_config(n, seed=1)
    cfg.flags |= abi.FLAG_LEAN_STEP  # as bench.py's rollout: only what the rollout reads is stored
    sim = BezSim(cfg, 0)
    act = (torch.rand(8, n * 18, device="cuda") * 2 - 1).contiguous()
    for t in range(24):  # summarize_profiles.py drops the first 4 launches per size
        sim.step(act[t % 8])
    torch.cuda.synchronize()
    sim.close()
print("done")
