#!/usr/bin/env python3
"""Time bez_ppo_wgrad_plan / _run (csrc/bez_wgrad.hip) on the bez_kickPPO.yaml shapes against the round-2 path (32-way batched GEMM + sum).
usage: python tools/wgrad_bench.py [all|L0|L1|L2|heads]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bez_isaacgym_amd.ppo import fused as F
dev = "cuda:0"
rows = 32768
ALL = {"L0": [(400, 54)], "L1": [(200, 400)], "L2": [(100, 200)], "heads": [(18, 100), (1, 100)], "all": [(400, 54), (200, 400), (100, 200), (18, 100), (1, 100)]}
for which in (sys.argv[1:] or ["all"]):
    shapes = ALL[which]
    dys = [torch.randn(rows, o, device=dev).half() for o, i in shapes]
    xs = [torch.randn(rows, i, device=dev).half() for o, i in shapes]
    grads = [torch.zeros(o, i, device=dev) for o, i in shapes]
    wg = F.WgradMfma(dys, xs, grads)
    ok = wg.ok and wg(True)
    if ok:
        for _ in range(5): wg(True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): wg(True)
        e1.record(); torch.cuda.synchronize()
        t_new = e0.elapsed_time(e1) * 1e3 / 50
    s = 32
    def old():
        for dy, x, g in zip(dys, xs, grads):
            part = torch.bmm(dy.view(s, rows // s, -1).transpose(1, 2), x.view(s, rows // s, -1))
            F.wgrad_sum(part, g, accumulate=True)
    for _ in range(5): old()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): old()
    e1.record(); torch.cuda.synchronize()
    print("%-6s wgrad_mfma %s us   bmm + wgrad_sum (round 2) %.1f us" % (which, ("%.1f" % t_new) if ok else "refused", e0.elapsed_time(e1) * 1e3 / 50))
