#!/usr/bin/env python3
"""Prices the "lane-group mapping" of SURVEY.md 7 on the articulated-body recursion of one six-joint chain (tools/lanegroup_probe.hip):
one env per lane (the product kernel's mapping) against eight lanes per env with DPP sums and ds_swizzle broadcasts.  Prints s_memtime
cycles per chain for both, the kernel times at 4096 envs, and checks that the two mappings agree.

    python tools/lanegroup_probe.py > profiles/r04_lanegroup_probe.txt
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
so = os.path.join(ROOT, "gpurun_out", "liblanegroup_probe.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + os.environ.get("PROBE_FLAGS", "").split() + ["-o", so,
                os.path.join(ROOT, "tools", "lanegroup_probe.hip")], check=True)   # PROBE_FLAGS=-fno-slp-vectorize: the product's build (no compiler packing)
lib = C.CDLL(so)
vp = C.c_void_p
lib.probe_run.argtypes = [C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_float, vp, vp, C.c_int, vp]
dev = "cuda:0"
torch.manual_seed(0)
n, nj = int(os.environ.get("N", 4096)), 6
# well-conditioned inputs: link inertias = SPD 6x6 of order 1e-2, joint axes of unit norm, small bias forces
m = torch.randn(nj, n, 6, 6, device=dev) * 0.05
LI = (m @ m.transpose(-1, -2) + 0.02 * torch.eye(6, device=dev)).contiguous()
pAl = (torch.randn(nj, n, 6, device=dev) * 0.1).contiguous()
S = torch.nn.functional.normalize(torch.randn(nj, n, 6, device=dev), dim=-1).contiguous()
cb = (torch.randn(nj, n, 6, device=dev) * 0.3).contiguous()
tau = (torch.randn(nj, n, device=dev) * 0.2).contiguous()
arm = 1e-3


def run(mapping, reps):
    out = torch.zeros(n, 7, device=dev)
    nwg = (n + 7) // 8 if mapping == 1 else ((n + 15) // 16 if mapping in (4, 5) else (n + 63) // 64)
    cyc = torch.zeros(nwg, dtype=torch.int64, device=dev)
    args = (mapping, vp(LI.data_ptr()), vp(pAl.data_ptr()), vp(S.data_ptr()), vp(cb.data_ptr()), vp(tau.data_ptr()), n, arm, vp(out.data_ptr()), vp(cyc.data_ptr()), reps, None)
    for _ in range(3):
        assert lib.probe_run(*args) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.probe_run(*args)
    e1.record(); torch.cuda.synchronize()
    return out, cyc.cpu().numpy() / reps, e0.elapsed_time(e1) * 1e3 / 20


# reference in fp64 (torch), the recursion as written in the .hip header
IA = torch.zeros(n, 6, 6, device=dev, dtype=torch.float64); pA = torch.zeros(n, 6, device=dev, dtype=torch.float64)
for j in range(nj - 1, -1, -1):
    IA = IA + LI[j].double(); pA = pA + pAl[j].double()
    s, c = S[j].double(), cb[j].double()
    U = (IA @ s.unsqueeze(-1)).squeeze(-1)
    D = (s * U).sum(-1) + arm
    uD = (tau[j].double() - (s * pA).sum(-1)) / D
    IA = IA - U.unsqueeze(-1) * U.unsqueeze(-2) / D.view(-1, 1, 1)
    pA = pA + (IA @ c.unsqueeze(-1)).squeeze(-1) + U * uD.unsqueeze(-1)
want = torch.cat([pA, IA.sum((-1, -2)).unsqueeze(-1)], -1)

REPS = 64
print("articulated-body recursion of one six-joint chain, %d envs, %d chains per wave-launch timed; extra compiler flags: %s" % (n, REPS, os.environ.get("PROBE_FLAGS", "(none: SLP vectoriser on)")))
res = {}
for mapping, name in ((0, "A  one lane per env (symmetric 6x6 in 21 registers)"), (1, "B  eight lanes per env (row per lane, DPP sums, ds_swizzle broadcasts)"),
                      (2, "A2 one lane per env, packed fp32 (full 6x6 as 18 register pairs, v_pk_fma_f32)"),
                      (3, "A3 one lane per env, column pairs + op_sel broadcasts (no moves, no horizontal adds)"),
                      (4, "C  four lanes per env, 8x8 as 2x2 blocks of 4x4, rank-1 update on v_mfma_f32_4x4x1_16B_f32"),
                      (5, "D  four lanes per env, 2x2 blocks of 3x3, quad_perm DPP only")):
    out, cyc, us = run(mapping, REPS)
    err = float((out.double() - want).abs().max() / want.abs().max())
    res[mapping] = (cyc, us)
    print("%-72s cycles per chain: median %6.0f  min %6.0f   kernel %.1f us for %d chains of all envs   max rel err vs fp64 %.1e" % (
        name, np.median(cyc), cyc.min(), us, REPS, err))
    assert err < 1e-4
a, b = np.median(res[0][0]), np.median(res[1][0])
print("latency of one chain: %.0f -> %.0f cycles = x%.2f; waves needed for 64 envs: 1 -> 8 (the idle three quarters of the chip at 4096 envs)" % (a, b, a / b))
print("chip-level throughput of the recursion alone at %d envs: %.1f -> %.1f us per %d chains" % (n, res[0][1], res[1][1], REPS))
c = np.median(res[2][0])
print("packed fp32 in the one-lane mapping: %.0f -> %.0f cycles per chain = x%.2f, same waves, same registers' worth of state" % (a, c, a / c))
c3 = np.median(res[3][0])
print("column pairs with op_sel broadcasts: %.0f -> %.0f cycles per chain = x%.2f (36 matrix registers instead of 21)" % (a, c3, a / c3))
c4 = np.median(res[4][0])
print("MFMA rank-1 update, four lanes per env: %.0f -> %.0f cycles per chain = x%.2f; waves needed for 64 envs: 1 -> 4; chip-level %.1f -> %.1f us per %d chains" % (
    a, c4, a / c4, res[0][1], res[4][1], REPS))
c5 = np.median(res[5][0])
print("quad blocks with quad_perm DPP, four lanes per env: %.0f -> %.0f cycles per chain = x%.2f; waves needed for 64 envs: 1 -> 4; chip-level %.1f -> %.1f us per %d chains" % (
    a, c5, a / c5, res[0][1], res[5][1], REPS))

# ---- the per-(env, joint) work of pass 2 that is NOT the recursion: one lane per env over six joints vs one lane per (env, joint)
lib.probe_joint_work.argtypes = [C.c_int, vp, C.c_int, vp, vp, C.c_int, vp]
qt = torch.nn.functional.normalize(torch.randn(nj, n, 4, device=dev), dim=-1)
links = torch.cat([qt, torch.randn(nj, n, 3, device=dev) * 0.1, torch.nn.functional.normalize(torch.randn(nj, n, 3, device=dev), dim=-1), torch.randn(nj, n, 6, device=dev),
                   torch.randn(nj, n, 3, device=dev) * 0.5, torch.rand(nj, n, 1, device=dev) * 0.2 + 0.02, torch.randn(nj, n, 3, device=dev) * 0.02,
                   torch.rand(nj, n, 3, device=dev) * 1e-4 + 1e-5], -1).contiguous()
assert links.shape[-1] == 26
jw = {}
for mode, name in ((0, "one lane per env, six joints one after the other"), (1, "one lane per (env, joint), eight lanes per env")):
    out = torch.zeros(n, device=dev)
    nwg = (n + 63) // 64 if mode == 0 else (n + 7) // 8
    cyc = torch.zeros(nwg, dtype=torch.int64, device=dev)
    args = (mode, vp(links.data_ptr()), n, vp(out.data_ptr()), vp(cyc.data_ptr()), REPS, None)
    for _ in range(3):
        assert lib.probe_joint_work(*args) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.probe_joint_work(*args)
    e1.record(); torch.cuda.synchronize()
    jw[mode] = (np.median(cyc.cpu().numpy() / REPS), e0.elapsed_time(e1) * 1e3 / 20, out.clone())
    print("per-joint packages (link inertia about the reference point, bias, drive / friction / limit terms) -- %-50s cycles per six joints: %6.0f   kernel %.1f us" % (name, jw[mode][0], jw[mode][1]))
err = float((jw[0][2] - jw[1][2]).abs().max() / jw[0][2].abs().max())
assert err < 1e-5, err
print("joint-per-lane: %.0f -> %.0f cycles for an env's six packages = x%.2f, eight times the waves (mappings agree to %.1e)" % (jw[0][0], jw[1][0], jw[0][0] / jw[1][0], err))
whole_now = a + jw[0][0]
for label, rec in (("MFMA recursion on 4 lanes per env", c4), ("DPP / swizzle recursion on 8 lanes per env (mapping B)", b)):
    whole_new = rec + jw[1][0]
    print("pass-2 window of one chain, recursion + packages: %.0f -> %.0f cycles = x%.2f (%s + packages one joint per lane; LDS hand-over of the packages not included)" % (
        whole_now, whole_new, whole_now / whole_new, label))
