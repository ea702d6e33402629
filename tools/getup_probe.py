#!/usr/bin/env python3
"""Get-up scenarios (SURVEY 8 f4 "get-up, kicks").

The reference ships three open-loop get-up tables (resources/library/trajectories/trajectories/simulation_getup{front,back,side}.csv,
numbers in tests/golden/trajectories.json) that `soccer_trajectories.py:56-91` plays through env.step at 0.00833 s of trajectory per
control step.  They are reference-held evidence about the rigid-body step that nothing else uses: they exercise drive authority under
load and upper-body / knee ground contact.  This probe lays the robot down (front: pitch +90 deg, the yaml's own "flat" quaternion
`bez_kick.yaml:20`; back: pitch -90 deg; side: roll +90 deg), lets it settle with all joints at 0 (the tables' first key frame),
plays the table through the SPLIT entry points (pre_physics + simulate: no fall reset interferes) and then holds the ready pose.

Back-ends: the CPU oracle (build container; experiment harness, never on the product path) or the HIP simulator through the C ABI
(`--backend hip`, GPU box).  Reported per scenario x variant: torso height and up-vector (z component of the torso's z axis) at the
end of the hold, their maxima during the motion, and the fraction of envs that end up standing (height > 0.28 m, up > 0.9).

    python tools/getup_probe.py --backend oracle --out profiles/r05_getup_oracle.json
    python tools/getup_probe.py --backend hip --envs 256 --out gpurun_out/r05_getup_hip.json
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bez_isaacgym_amd import abi  # noqa: E402
from tests.scenarios import STARTS, VARIANTS, lay_down, make_backend, play  # noqa: E402

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="oracle", choices=("oracle", "hip"))
    ap.add_argument("--envs", type=int, default=16)
    ap.add_argument("--out", default=None)
    ap.add_argument("--flags", type=int, default=None, help="flag word (oracle-only model variants, e.g. the full ground-shape set)")
    ap.add_argument("--scenarios", nargs="*", default=list(STARTS))
    ap.add_argument("--variants", nargs="*", default=list(VARIANTS))
    ap.add_argument("--trace", type=int, default=None)
    a = ap.parse_args()
    model = json.load(open(os.path.join(ROOT, "bez_isaacgym_amd", "model", "bez_model.json")))
    out = {"backend": a.backend, "envs": a.envs, "flags": a.flags, "rows": []}
    print("%-11s %-26s | final z   up  | max z   up  | standing" % ("scenario", "variant"))
    for name in a.scenarios:
        for vname in a.variants:
            cfg = abi.default_config(a.envs, seed=7)
            for k, v in VARIANTS[vname].items():
                setattr(cfg, k, v)
            if a.flags is not None:
                cfg.flags = a.flags
            sim = make_backend(a.backend, cfg)
            sim.step(np.zeros((a.envs, 18), np.float32))  # first call: reset of every env (reset_buf starts at 1)
            lay_down(sim, a.envs, name, np.random.default_rng(3))
            r = play(sim, a.envs, name, model, trace_env=a.trace)
            r.update(scenario=name, variant=vname)
            out["rows"].append(r)
            print("%-11s %-26s | %6.3f %5.2f | %6.3f %5.2f | %5.2f" % (name, vname, r["final_z"], r["final_up"], r["max_z"], r["max_up"], r["standing"]), flush=True)
            if a.trace is not None:
                for row in r["trace"]:
                    print("    step %4d  z %.3f  up %.2f" % row)
            del sim
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
