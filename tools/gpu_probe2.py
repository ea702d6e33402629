#!/usr/bin/env python3
"""Per-env error distribution HIP / fp32 oracle vs the fp64 oracle after one resynchronised control step (round 6: separates the general
error level -- percentiles -- from branch flips -- the tail)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bez_isaacgym_amd import abi
from oracle.bez_oracle import Oracle
from tests.sim_adapter import SimAdapter

n = int(os.environ.get("N", 512)); steps = int(os.environ.get("STEPS", 40))
o, g = Oracle(abi.default_config(n, seed=7)), SimAdapter(abi.default_config(n, seed=7))
o32 = Oracle(abi.default_config(n, seed=7), precision="f32")
rng = np.random.default_rng(3)
E = {}
for t in range(steps):
    for x in (g, o32):
        x.set_root_states(o.root_states); x.set_dof_state(o.dof_state); x.set_contact_forces(o.contact_forces)
        x.set_targets(o.targets); x.set_reset(o.reset_buf); x.set_progress(o.progress_buf)
    act = rng.uniform(-1, 1, (n, 18)).astype(np.float32)
    o.step(act); g.step(act); o32.step(act)
    for tag, x in (("hip", g), ("cpu32", o32)):
        ro, rx = o.root_states.reshape(n, 2, 13), x.root_states.reshape(n, 2, 13)
        do, dx = o.dof_state.reshape(n, 18, 2), x.dof_state.reshape(n, 18, 2)
        for k, a, b in (("root pose", ro[:, 0, :7], rx[:, 0, :7]), ("root vel", ro[:, 0, 7:], rx[:, 0, 7:]), ("q", do[:, :, 0], dx[:, :, 0]), ("qd", do[:, :, 1], dx[:, :, 1]),
                        ("cf", o.contact_forces.reshape(n, -1), x.contact_forces.reshape(n, -1)), ("rew", o.rew[:, None], x.rew[:, None])):
            E.setdefault((tag, k), []).append(np.abs(a - b).max(1))
print("%-18s %10s %10s %10s %10s %10s" % ("", "median", "p99", "p99.9", "max", "n>10*p99"))
for (tag, k), v in sorted(E.items()):
    v = np.concatenate(v)
    p99 = np.quantile(v, 0.99)
    print("%-6s %-11s %10.2e %10.2e %10.2e %10.2e %10d" % (tag, k, np.median(v), p99, np.quantile(v, 0.999), v.max(), int((v > 10 * max(p99, 1e-9)).sum())))
