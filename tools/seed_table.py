#!/usr/bin/env python3
"""One table over seeds of what a default-yaml training ends as (the trained reward lands near 19 or near 30).

For every seed: `python -m bez_isaacgym_amd.train task=bez_kick num_envs=4096 headless=True max_iterations=E seed=S` with a checkpoint at
the last epoch, then that checkpoint played deterministically in 4096 fresh envs for 900 steps (tools/sim2sim_gpu.evaluate): trained
reward (mean of the last 2000 logged epochs), goal rate, episode length, goal episodes' length and which test ends the other episodes
(fall / out of bounds / goal-angle / timeout; kick_env.py:1198-1395).  GPU box.

    python tools/seed_table.py --seeds 42 43 44 45 46 47 48 49 --out profiles/r06_seed_table.txt
"""
import argparse
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
LINE = re.compile(r"^epoch (\d+) frames (\d+) fps total (\d+).* mean_reward (-?[\d.]+)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="+", default=[42, 43, 44, 45, 46, 47, 48, 49])
    ap.add_argument("--epochs", type=int, default=6156)
    ap.add_argument("--out", default="gpurun_out/seed_table.txt")
    ap.add_argument("overrides", nargs="*")
    a = ap.parse_args()
    rows = []
    for seed in a.seeds:
        t0 = time.time()
        shutil.rmtree(os.path.join(ROOT, "runs"), ignore_errors=True)
        cmd = [sys.executable, "-m", "bez_isaacgym_amd.train", "task=bez_kick", "num_envs=4096", "headless=True", "max_iterations=%d" % a.epochs,
               "seed=%d" % seed, "train.params.config.save_frequency=%d" % a.epochs, "train.params.config.save_best_after=1000000000"] + a.overrides
        p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True)
        rew = [float(m.group(4)) for m in (LINE.match(ln) for ln in p.stdout.splitlines()) if m]
        fps = [int(m.group(3)) for m in (LINE.match(ln) for ln in p.stdout.splitlines()) if m]
        ck = sorted(glob.glob(os.path.join(ROOT, "runs", "*", "nn", "last_*_ep_%d.pth" % a.epochs)))
        row = dict(seed=seed, trained=sum(rew[-2000:]) / max(len(rew[-2000:]), 1) if rew else float("nan"), fps=fps[-1] if fps else 0, rc=p.returncode)
        if ck:
            ev = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sim2sim_gpu.py"), "--checkpoint", ck[-1], "--envs", "4096", "--steps", "900"],
                                cwd=ROOT, capture_output=True, text=True)
            try:
                r = json.loads(ev.stdout[ev.stdout.index("{"):])
                row.update(goal_rate=r["goal_rate"], mean_length=r["mean_length"], goal_length=r["goal_length"], mean_return=r["mean_return"], reasons=r["reasons"], episodes=r["episodes"])
            except Exception as ex:   # noqa: BLE001
                row["eval_error"] = str(ex) + ev.stderr[-500:]
        else:
            row["eval_error"] = "no checkpoint: " + p.stderr[-500:]
        row["seconds"] = round(time.time() - t0, 1)
        rows.append(row)
        print(json.dumps(row), flush=True)
    with open(a.out, "w") as f:
        f.write("python tools/seed_table.py --seeds %s --epochs %d %s\n" % (" ".join(map(str, a.seeds)), a.epochs, " ".join(a.overrides)))
        f.write("%-5s %8s %9s %9s %9s %9s   %s\n" % ("seed", "trained", "goal rate", "ep length", "goal len", "return", "episodes ended by goal / fall / out of bounds / goal angle / timeout (share)"))
        for r in rows:
            if "reasons" in r:
                e = max(r["episodes"], 1); q = r["reasons"]
                f.write("%-5d %8.2f %9.3f %9.1f %9.1f %9.2f   %.3f / %.3f / %.3f / %.3f / %.3f\n" % (r["seed"], r["trained"], r["goal_rate"], r["mean_length"], r["goal_length"], r["mean_return"],
                                                                                              q["goal"] / e, q["fall"] / e, q["oob"] / e, q["angle"] / e, q["timeout"] / e))
            else:
                f.write("%-5d %8.2f   evaluation failed: %s\n" % (r["seed"], r["trained"], r.get("eval_error", "")[:200]))
    shutil.rmtree(os.path.join(ROOT, "runs"), ignore_errors=True)


if __name__ == "__main__":
    main()
