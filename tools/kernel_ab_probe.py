#!/usr/bin/env python3
"""Worst per-step differences between the fused-step kernel variants (tolerances of tests/test_gpu_round2.py::test_fused_step_kernels_agree)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from tests.sim_adapter import SimAdapter
from tests.test_tasks import make_cfg
n = 200
for variant in ("kick", "kick_cleats", "walk", "orient", "orient_cleats"):
    for other in ("lane",):
        task = "bez_walk" if variant.startswith("walk") else ("bez_orient" if variant.startswith("orient") else "bez_kick")
        kw = dict(seed=31, task=task, cleats=variant.endswith("_cleats"))
        os.environ.pop("BEZ_SIM_KERNEL", None)
        a = SimAdapter(make_cfg(n, **kw))
        os.environ["BEZ_SIM_KERNEL"] = other
        b = SimAdapter(make_cfg(n, **kw))
        os.environ.pop("BEZ_SIM_KERNEL", None)
        rng = np.random.default_rng(8)
        w = {}
        def upd(k, x, y):
            w[k] = max(w.get(k, 0.0), float(np.max(np.abs(np.asarray(x, np.float64) - np.asarray(y, np.float64)))))
        for t in range(40):
            b.set_root_states(a.root_states); b.set_dof_state(a.dof_state); b.set_contact_forces(a.contact_forces)
            b.set_targets(a.targets); b.set_reset(a.reset_buf); b.set_progress(a.progress_buf); b.set_prev_lin_vel(a.prev_lin_vel)
            if task != "bez_kick": b.set_goal(a.goal)
            act = rng.uniform(-1, 1, (n, 18)).astype(np.float32)
            a.step(act); b.step(act)
            ra, rb = a.root_states, b.root_states
            upd("root_pos_quat", rb[..., 0:7], ra[..., 0:7]); upd("root_vel", rb[..., 7:13], ra[..., 7:13])
            da, db = a.dof_state.reshape(n, 18, 2), b.dof_state.reshape(n, 18, 2)
            upd("q", db[..., 0], da[..., 0]); upd("qd", db[..., 1], da[..., 1])
            ca, cb = a.contact_forces, b.contact_forces
            upd("cf_abs", cb, ca); w["cf_rel"] = max(w.get("cf_rel", 0), float(np.max(np.abs(cb - ca) / (np.abs(ca) + 1.0))))
            upd("obs_imu", b.obs[:, 36:42], a.obs[:, 36:42]); upd("obs_orn", b.obs[:, 42:44], a.obs[:, 42:44])
            w["feet_eq"] = min(w.get("feet_eq", 1.0), float(np.mean(b.obs[:, 44:52] == a.obs[:, 44:52])))
            upd("rew", b.rew, a.rew)
            w["int_mismatch"] = w.get("int_mismatch", 0) + int((b.reset_buf != a.reset_buf).sum() + (b.progress_buf != a.progress_buf).sum())
        print(variant, other, {k: (round(v, 8) if isinstance(v, float) else v) for k, v in w.items()})
