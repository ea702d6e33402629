#!/usr/bin/env python3
"""GPU debug helper: resynchronised single-step parity HIP vs fp64 oracle, printing the worst env's joint rates when a bar is exceeded.
    BEZ_SIM_KERNEL=lane|ws8 python tools/parity_debug.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bez_isaacgym_amd import abi
from oracle.bez_oracle import Oracle
from tests.sim_adapter import SimAdapter

n = 128
o = Oracle(abi.default_config(n, seed=7)); g = SimAdapter(abi.default_config(n, seed=7))
rng = np.random.default_rng(3)
np.set_printoptions(precision=3, suppress=True, linewidth=220)
shown = 0
for t in range(40):
    g.set_root_states(o.root_states); g.set_dof_state(o.dof_state); g.set_contact_forces(o.contact_forces); g.set_targets(o.targets)
    g.set_reset(o.reset_buf); g.set_progress(o.progress_buf)
    q0 = o.dof_state.reshape(n, 18, 2).copy()
    act = rng.uniform(-1, 1, (n, 18)).astype(np.float32)
    o.step(act); g.step(act)
    do, dg = o.dof_state.reshape(n, 18, 2), g.dof_state.reshape(n, 18, 2)
    e = np.abs(do[:, :, 1] - dg[:, :, 1]).max(1)
    r = np.abs(o.root_states.reshape(n, 2, 13)[:, 0] - g.root_states.reshape(n, 2, 13)[:, 0]).max(1)
    bad = np.where((e > 1.5e-2) | (r > 4e-3))[0]
    rst = o.reset_buf
    for i in bad[:3]:
        if shown < 8:
            shown += 1
            print("step", t, "env", i, "kernel", os.environ.get("BEZ_SIM_KERNEL", "ws8"), "dof-vel err %.3g root err %.3g reset %d" % (e[i], r[i], rst[i]))
            print("  qd start ", q0[i, :, 1]); print("  qd oracle", do[i, :, 1]); print("  qd hip   ", dg[i, :, 1])
            print("  q  diff  ", do[i, :, 0] - dg[i, :, 0])
            print("  cf oracle sum", np.abs(o.contact_forces.reshape(n, 22, 3)[i]).sum(0), "hip", np.abs(g.contact_forces.reshape(n, 22, 3)[i]).sum(0))
print("done; worst dof-vel err over the run: see above" if shown else "all 40 steps within the bars")
