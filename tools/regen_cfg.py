#!/usr/bin/env python3
"""Regenerates bez_isaacgym_amd/cfg/*.yaml from the reference's config VALUES (build container only).
Configs are data that must stay drop-in (same keys / defaults / interpolations); this re-serialises
them (sorted keys, flow lists, our own header comments) rather than copying the files."""
import os
import yaml

REF = os.environ.get("BEZ_REFERENCE_ROOT", "/root/reference") + "/bez_isaacgym/cfg"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bez_isaacgym_amd", "cfg")
FILES = ["config.yaml", "task/bez_kick.yaml", "task/bez_kick_test.yaml", "train/bez_kickPPO.yaml",
         "task/bez_walk.yaml", "task/bez_orient.yaml", "train/bez_walkPPO.yaml", "train/bez_orientPPO.yaml"]

if __name__ == "__main__":
    for rel in FILES:
        dst = os.path.join(OUT, rel)
        header = "".join(l for l in open(dst) if l.startswith("#")) if os.path.exists(dst) else \
            "# %s (values: reference cfg/%s)\n" % (os.path.basename(rel), rel)
        data = yaml.safe_load(open(os.path.join(REF, rel)))
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        with open(dst, "w") as f:
            f.write(header)
            yaml.safe_dump(data, f, default_flow_style=None, sort_keys=True, width=120)
        print("wrote", dst)
