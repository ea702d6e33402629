#!/usr/bin/env python3
"""Build-container only: key-frame NUMBERS of the reference's scripted trajectories
(resources/library/trajectories/trajectories/simulation_*.csv) -> tests/golden/trajectories.json (data only; comments dropped)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bez_isaacgym_amd.utils.trajectories import read_csv_table  # noqa: E402

SRC = "/root/reference/resources/library/trajectories/trajectories"
out = {}
for name in ("simulation_rightkick", "simulation_getupfront", "simulation_getupback", "simulation_getupside"):
    out[name] = read_csv_table(os.path.join(SRC, name + ".csv"))
with open(os.path.join(ROOT, "tests", "golden", "trajectories.json"), "w") as f:
    json.dump(out, f, indent=0)
print({k: (len(v["time"]), len(v["joints"])) for k, v in out.items()})
