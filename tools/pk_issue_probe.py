#!/usr/bin/env python3
"""Cycles per instruction of one wave's stream of independent v_fma_f32 / v_pk_fma_f32 (tools/pk_issue_probe.hip)."""
import ctypes as C, os, subprocess
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
so = os.path.join(ROOT, "gpurun_out", "libpk_issue_probe.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tools", "pk_issue_probe.hip")], check=True)
lib = C.CDLL(so)
lib.issue_run.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
iters, blocks = 1000, 64
for pk, name in ((0, "v_fma_f32"), (1, "v_pk_fma_f32")):
    out = torch.zeros(blocks * 64, device="cuda"); cyc = torch.zeros(blocks, dtype=torch.int64, device="cuda")
    for _ in range(2):
        assert lib.issue_run(pk, C.c_void_p(out.data_ptr()), C.c_void_p(cyc.data_ptr()), iters, blocks) == 0
    torch.cuda.synchronize()
    per = cyc.cpu().numpy() / (iters * 64.0)
    print("%-14s %.2f cycles per instruction (one wave per CU, 64 independent instructions per loop pass)%s" % (name, per.mean(), "  = %.2f per fp32 FMA" % (per.mean() / 2) if pk else ""))
