#!/usr/bin/env python3
"""where ws8q and ws8 part: per step, the envs whose joint rates differ by more than 0.05 rad/s (resynchronised every step), and what the
fp64 oracle says about them (its joint rates, the margin of its closest speed-limit decision).  A diagnostic beside the parity tests
(it uses the oracle, hence its place under tests/):   python tests/ws8q_debug.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from tests.sim_adapter import SimAdapter
from tests.test_tasks import make_cfg
n = 200
os.environ["BEZ_SIM_KERNEL"] = "ws8"; a = SimAdapter(make_cfg(n, seed=31, task="bez_kick"))
os.environ["BEZ_SIM_KERNEL"] = "ws8q"; b = SimAdapter(make_cfg(n, seed=31, task="bez_kick"))
from oracle.bez_oracle import Oracle
o = Oracle(make_cfg(n, seed=31, task="bez_kick"), precision="f64")
rng = np.random.default_rng(8)
for t in range(40):
    pre = dict(root=a.root_states, dof=a.dof_state, cf=a.contact_forces, tg=a.targets, rs=a.reset_buf, pg=a.progress_buf, pv=a.prev_lin_vel)
    b.set_root_states(a.root_states); b.set_dof_state(a.dof_state); b.set_contact_forces(a.contact_forces)
    b.set_targets(a.targets); b.set_reset(a.reset_buf); b.set_progress(a.progress_buf); b.set_prev_lin_vel(a.prev_lin_vel)
    pre_reset = a.reset_buf.copy()
    act = rng.uniform(-1, 1, (n, 18)).astype(np.float32)
    a.step(act); b.step(act)
    da, db = a.dof_state.reshape(n, 18, 2), b.dof_state.reshape(n, 18, 2)
    e = np.abs(da[..., 1] - db[..., 1]).max(1)
    bad = np.nonzero(e > 0.05)[0]
    if len(bad):
        cf = a.contact_forces.reshape(n, -1, 3)
        print("step", t, "envs", bad.tolist(), "err", e[bad].round(3).tolist(), "reset-before", pre_reset[bad].tolist(), "joint", np.abs(da[bad, :, 1] - db[bad, :, 1]).argmax(1).tolist(),
              "ball cf", np.abs(cf[bad, -1]).max(1).round(2).tolist(), flush=True)
        o.set_root_states(pre["root"]); o.set_dof_state(pre["dof"]); o.set_contact_forces(pre["cf"]); o.set_targets(pre["tg"])
        o.set_reset(pre["rs"]); o.set_progress(pre["pg"]); o.set_prev_lin_vel(pre["pv"])
        o.step(act)
        do = o.dof_state.reshape(n, 18, 2)
        for e_ in bad:
            j = int(np.abs(da[e_, :, 1] - db[e_, :, 1]).argmax())
            print("   env", e_, "joint", j, "qd: ws8 %.4f  ws8q %.4f  oracle f64 %.4f   oracle's margin of the closest speed-limit decision %.2e rad/s; all-joint max |ws8 - oracle| %.3g, |ws8q - oracle| %.3g" % (
                da[e_, j, 1], db[e_, j, 1], do[e_, j, 1], o.vlim_margin[e_], np.abs(da[e_, :, 1] - do[e_, :, 1]).max(), np.abs(db[e_, :, 1] - do[e_, :, 1]).max()), flush=True)
print("done")
