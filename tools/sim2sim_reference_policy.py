#!/usr/bin/env python3
"""Build-container-only experiment: run the reference's shipped policy (Bez_Kick_33.pth, read WITHOUT unpickling) deterministically
(mu, as utils/players.py does) in this build's CPU oracle and report what it achieves.  A policy trained against PhysX contact is not
expected to transfer unchanged to a different contact model; this is evidence, not a pass/fail gate."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bez_isaacgym_amd import abi
from bez_isaacgym_amd.ppo.a2c_continuous import ModelA2CContinuousLogStd, RunningMeanStd
from bez_isaacgym_amd.utils.rlg_checkpoint import load_into_agent_modules, read_rlgames_checkpoint
from oracle.bez_oracle import Oracle

ck = read_rlgames_checkpoint("/root/reference/bez_isaacgym/results/Bez_Kick/Normal/Bez_Kick_33.pth")
m = ModelA2CContinuousLogStd(54, 18, (400, 200, 100)).eval(); rms = RunningMeanStd((54,)).eval()
load_into_agent_modules(ck, m, rms)
n = 256
o = Oracle(abi.default_config(n, seed=1))
o.step(np.zeros((n, 18), np.float32))
ep_ret = np.zeros(n); finished = []; lengths = []; cur_len = np.zeros(n); reasons = {"goal": 0, "fall/oob/angle": 0, "timeout": 0}
for t in range(1200):
    with torch.no_grad():
        mu, _, _ = m.a2c_network(rms(torch.from_numpy(o.obs)))
    o.step(np.clip(mu.numpy(), -1.0, 1.0).astype(np.float32))
    r, d = o.rew, o.reset_buf
    ep_ret += r; cur_len += 1
    for i in np.where(d == 1)[0]:
        finished.append(ep_ret[i]); lengths.append(cur_len[i])
        reasons["goal" if r[i] > 1.0 else ("timeout" if o.progress_buf[i] >= 900 else "fall/oob/angle")] += 1
        ep_ret[i] = 0; cur_len[i] = 0
print("episodes finished:", len(finished), "mean return %.2f" % np.mean(finished), "median %.2f" % np.median(finished),
      "mean length %.1f" % np.mean(lengths), reasons)
