#!/usr/bin/env python3
"""Writes tests/golden/urdf_bodies.json: the NUMBERS of the reference's robot description
(resources/assets/bez/model/soccerbot_stl.urdf: per-body mass / COM / inertia, joint parent / origin / axis / limits), read by
the independent parser of tests/urdf_independent.py -- not by bez_isaacgym_amd/model/compile_model.py, whose output this
fixture exists to cross-check (tests/test_model_independent.py).  Run in the build container only (needs /root/reference)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.urdf_independent import FIXTURE, URDF_REL, parse_urdf  # noqa: E402

if __name__ == "__main__":
    ref = os.environ.get("BEZ_REFERENCE_ROOT", "/root/reference")
    bodies = parse_urdf(os.path.join(ref, URDF_REL))
    with open(FIXTURE, "w") as f:
        json.dump(dict(source=URDF_REL, bodies=bodies), f, indent=1)
    print(FIXTURE, len(bodies), "bodies")
