#!/usr/bin/env python3
"""Diagnostic: where the one-launch policy forward (csrc/bez_policy.hip) spends its time.  Builds bez_policy.hip with -DBEZ_PF_STAMPS,
runs the forward on 4096 x 54 -> 400 -> 200 -> 100 -> (18 + 1) and prints s_memtime of workgroup 0 / thread 0 at the phase boundaries."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
SRC = os.environ.get("PF_SRC", os.path.join(ROOT, "bez_isaacgym_amd", "csrc", "bez_policy.hip"))   # (another tree's copy for an A/B)
so = os.path.join(ROOT, "gpurun_out", "libbez_policy_stamps.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize", "-DBEZ_PF_STAMPS", "-o", so,
                SRC], check=True)
lib = C.CDLL(so)
dev = "cuda:0"
torch.manual_seed(0)
n, d, units, a = int(os.environ.get("N", 4096)), 54, (400, 200, 100), 18
dims = [d] + list(units)
hid = [((torch.randn(dims[i + 1], dims[i], device=dev) / dims[i] ** 0.5).half().contiguous(), (torch.randn(dims[i + 1], device=dev) * 0.1).half()) for i in range(3)]
mu_w, mu_b = (torch.randn(a, 100, device=dev) / 10).half().contiguous(), torch.zeros(a, device=dev).half()
v_w, v_b = (torch.randn(1, 100, device=dev) / 10).half().contiguous(), torch.zeros(1, device=dev).half()
obs = torch.randn(n, d, device=dev)
mu, val = torch.empty(n, a, device=dev), torch.empty(n, 1, device=dev)
stamps = torch.zeros(16 + 64 * 6, dtype=torch.int64, device=dev)
lib.bez_ppo_policy_debug_stamps(C.c_void_p(stamps.data_ptr()))
hw = (C.c_void_p * 3)(*[w.data_ptr() for w, _ in hid]); hb = (C.c_void_p * 3)(*[b.data_ptr() for _, b in hid]); wd = (C.c_int32 * 3)(*units)
vp = C.c_void_p
lib.bez_ppo_policy_forward.argtypes = [vp, C.c_int64, C.c_int32, vp, vp, C.c_float, C.c_int32, vp, vp, vp, vp, vp, C.c_int32, vp, vp, vp, vp, C.c_int32, vp]
def run():
    return lib.bez_ppo_policy_forward(vp(obs.data_ptr()), n, d, None, None, 0.0, 3, C.cast(hw, vp), C.cast(hb, vp), C.cast(wd, vp), vp(mu_w.data_ptr()), vp(mu_b.data_ptr()),
                                      a, vp(v_w.data_ptr()), vp(v_b.data_ptr()), vp(mu.data_ptr()), vp(val.data_ptr()), 0, None)  # row-major weights (weights_packed = 0)
PACKED = int(os.environ.get("PACKED", "0"))
if PACKED:  # fragment-major copies (the layout of include/bez_sim.h, weights_packed), built with the product's own index helper
    import numpy as np
    from bez_isaacgym_amd.ppo.fused import _pack_index
    def pack(w):
        idx, size = _pack_index(*w.shape)
        out = torch.zeros(size, device=dev, dtype=torch.float16)
        out[torch.from_numpy(idx.reshape(-1)).to(dev)] = w.reshape(-1)
        return out
    pk = [pack(w) for w, _ in hid]
    heads_pk = pack(torch.cat([mu_w, v_w], 0))
    hw = (C.c_void_p * 3)(*[w.data_ptr() for w in pk])
    def run():
        return lib.bez_ppo_policy_forward(vp(obs.data_ptr()), n, d, None, None, 0.0, 3, C.cast(hw, vp), C.cast(hb, vp), C.cast(wd, vp), vp(heads_pk.data_ptr()),
                                          vp(mu_b.data_ptr()), a, vp(v_w.data_ptr()), vp(v_b.data_ptr()), vp(mu.data_ptr()), vp(val.data_ptr()), 1, None)
if os.environ.get("ROLL"):   # the rollout-step form of the launch (sampling epilogue + the previous step's bookkeeping), contiguous rows
    assert PACKED
    f = lambda *sh: torch.zeros(*sh, device=dev)
    logstd, noise, dones = f(a), torch.randn(n, a, device=dev), f(n)
    mb_obs, mb_dones, mb_mu, mb_val, act, act_env, nlp, sig = f(n, d), f(n), f(n, a), f(n), f(n, a), f(n, a), f(n), f(n, a)
    class Post(C.Structure):
        _fields_ = [("rew", vp), ("reset", vp), ("timeouts", vp), ("prev_values", vp), ("reward_scale", C.c_float), ("gamma", C.c_float), ("bootstrap", C.c_int32),
                    ("shaped", vp), ("dones_f", vp), ("cur_rew", vp), ("cur_len", vp), ("ep_stats", vp), ("ep_parts", vp)]
    keep = [f(n), torch.zeros(n, dtype=torch.int64, device=dev), torch.zeros(n, dtype=torch.int64, device=dev), f(n), f(n), f(n), f(n), f(n),
            torch.zeros(3, dtype=torch.float64, device=dev)]
    post = Post(keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), keep[3].data_ptr(), 0.01, 0.99, 1, keep[4].data_ptr(), keep[5].data_ptr(),
                keep[6].data_ptr(), keep[7].data_ptr(), keep[8].data_ptr(), None)
    P = lambda t: vp(t.data_ptr())
    def run():
        return lib.bez_ppo_policy_rollout_step(P(obs), C.c_int64(n), C.c_int32(d), None, None, C.c_float(0.0), C.c_int32(3), C.cast(hw, vp), C.cast(hb, vp), C.cast(wd, vp),
                                               P(heads_pk), P(mu_b), C.c_int32(a), P(v_w), P(v_b), P(logstd), P(noise), P(dones), None, None, C.c_float(0.0),
                                               P(mb_obs), P(mb_dones), P(mb_mu), P(mb_val), P(act), P(act_env), P(nlp), P(sig), C.c_int32(1),
                                               C.byref(post) if os.environ["ROLL"] == "2" else None, None, None, None, None)
if os.environ.get("TRAIN"):   # the training forward (activations kept for the backward pass): 8 waves per 64-row tile, two workgroups per CU
    assert PACKED
    x0 = torch.empty(n, d, device=dev, dtype=torch.float16)
    acts = [torch.empty(n, u, device=dev, dtype=torch.float16) for u in units]
    ao = (C.c_void_p * 3)(*[t.data_ptr() for t in acts])
    def run():
        return lib.bez_ppo_policy_forward_train(vp(obs.data_ptr()), C.c_int64(n), C.c_int32(d), None, None, C.c_float(0.0), C.c_int32(3), C.cast(hw, vp), C.cast(hb, vp),
                                                C.cast(wd, vp), vp(heads_pk.data_ptr()), vp(mu_b.data_ptr()), C.c_int32(a), vp(v_w.data_ptr()), vp(v_b.data_ptr()),
                                                vp(x0.data_ptr()), C.cast(ao, vp), vp(mu.data_ptr()), vp(val.data_ptr()), C.c_int32(1), None)
for _ in range(20):
    assert run() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    run()
e1.record(); torch.cuda.synchronize()
print("kernel %.2f us per launch (n = %d, weights %s)" % (e0.elapsed_time(e1) * 1e3 / 200, n, "fragment-major" if PACKED else "row-major"))
s = stamps.cpu().numpy()
names = {0: "entry", 1: "obs staged + sync", 2: "layer 0 done (wave 0)", 3: "after sync", 4: "layer 1 done", 5: "after sync", 6: "layer 2 done", 7: "after sync", 13: "epilogue: sampled + sync", 12: "epilogue: per-row done", 14: "heads done", 15: "end"}
for k in sorted(names, key=lambda k: s[k]):
    if s[k]:
        print("%-24s %8d" % (names[k], s[k] - s[0]))
print("per wave (relative to entry): layer, wave: product start / product done / epilogue done [/ second block done]")
for L in range(3):
    for w in range(16):
        v = s[16 + L * 64 + w * 4: 16 + L * 64 + w * 4 + 4]
        if v[0]:
            print("  L%d w%d  %s" % (L, w, "  ".join("%7d" % (x - s[0]) if x else "      -" for x in v)))
