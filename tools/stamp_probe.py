#!/usr/bin/env python3
"""Diagnostic: builds libbez_sim_stamps.so with -DBEZ_WS_STAMPS and prints where workgroup 0's four role waves spend their
cycles (s_memtime at phase boundaries).  Never used for timing claims (the stamps perturb the kernel)."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
from bez_isaacgym_amd import abi
so = os.path.join(ROOT, "gpurun_out", "libbez_sim_stamps.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
import glob
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize", "-DBEZ_WS_STAMPS", "-o", so] +
               sorted(glob.glob(os.path.join(ROOT, "bez_isaacgym_amd", "csrc", "*.hip"))), check=True)
lib = C.CDLL(so)
cfg = abi.default_config(4096)
h = C.c_void_p()
assert lib.bez_sim_create(C.byref(cfg), 0, C.byref(h)) == 0
acts = (torch.rand(40, 4096 * 18, device="cuda") * 2 - 1).contiguous()
for t in range(30):
    lib.bez_sim_step(h, C.c_void_p(acts[t].data_ptr()), None)
NR = 8
out = (C.c_ulonglong * (NR * 32))()
allrows = []
for t in range(5):
    assert lib.bez_sim_debug_stamps(h, C.c_void_p(acts[30 + t].data_ptr()), out) == 0
    allrows.append(np.array(list(out), dtype=np.int64).reshape(NR, 32))
a = allrows[-1]
names = {18: "actions staged", 22: "kernel entry", 0: "loads done", 1: "after B0", 2: "s0 pass1 done", 3: "s0 after B1", 4: "s0 pass2/solve done", 5: "s0 after B2", 6: "s0 after B3",
         7: "s0 pass3 done", 8: "s0 after B4", 9: "s0 after B5", 10: "s1 pass1 done", 11: "s1 after B1", 12: "s1 pass2/solve done", 13: "s1 after B2",
         14: "s1 after B3", 15: "s1 pass3 done", 16: "s1 after B4", 17: "s1 after B5", 26: "pass2 joint 0 done", 27: "pass2 joint 1 done", 28: "pass2 joint 2 done", 29: "pass2 joint 3 done", 30: "pass2 joint 4 done", 31: "pass2 joint 5 (foot) done", 19: "root stores done", 21: "post done (before last B5; root: reward done)", 23: "kernel end", 24: "s0 before B1c", 25: "s1 before B1c"}
order = [22, 18, 0, 1, 2, 3, 24, 4, 5, 6, 7, 8, 9, 10, 11, 31, 30, 29, 28, 27, 26, 25, 12, 13, 14, 15, 16, 19, 21, 17, 23]
t0 = a[:, 22][a[:, 22] > 0].min()
roles = ["L-leg", "R-leg", "upper", "root", "r4", "r5", "r6", "r7"]
print(("%-22s" + " %8s" * NR + "   (cycles since first wave entry; s_memtime ticks; kernel = %s)") % ("phase", *roles, os.environ.get("BEZ_SIM_KERNEL", "ws")))
for k in order:
    row = [(a[r, k] - t0) if a[r, k] else -1 for r in range(NR)]
    print(("%-22s" + " %8d" * NR) % (names[k], *row))
