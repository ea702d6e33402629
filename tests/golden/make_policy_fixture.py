#!/usr/bin/env python3
"""Build-container only: write tests/golden/bez_kick_33_policy.npz -- NUMBERS ONLY -- from the reference's shipped
rl_games checkpoint (bez_isaacgym/results/Bez_Kick/Normal/Bez_Kick_33.pth), read WITHOUT unpickling it
(bez_isaacgym_amd/utils/rlg_checkpoint.py).  Contents: the actor-critic MLP weights under their rl_games key names,
the 54-d observation running mean / variance / count, the value normaliser, and the scalar training facts
(epoch, frame, last_mean_rewards).  The GPU box has no /root/reference: tests there use this fixture."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bez_isaacgym_amd.utils.rlg_checkpoint import read_rlgames_checkpoint  # noqa: E402

SRC = "/root/reference/bez_isaacgym/results/Bez_Kick/Normal/Bez_Kick_33.pth"
OUT = os.path.join(ROOT, "tests", "golden", "bez_kick_33_policy.npz")


def main():
    ck = read_rlgames_checkpoint(SRC)
    out = {}
    for k, v in ck["model"].items():
        if isinstance(v, np.ndarray):
            out["model/" + k] = v.astype(np.float32)
    for grp in ("running_mean_std", "reward_mean_std"):
        for k in ("running_mean", "running_var", "count"):
            out[grp + "/" + k] = np.asarray(ck[grp][k], dtype=np.float64)
    out["epoch"] = np.int64(ck["epoch"]); out["frame"] = np.int64(ck["frame"])
    out["last_mean_rewards"] = np.float64(ck["last_mean_rewards"])
    np.savez_compressed(OUT, **out)
    n = sum(v.size for k, v in out.items() if k.startswith("model/"))
    print("wrote", OUT, "params", n, "bytes", os.path.getsize(OUT))


if __name__ == "__main__":
    main()
