#!/usr/bin/env python3
"""Learning curves at the reference checkpoint's length (6156 epochs = 806 879 232 frames, results/Bez_Kick/Normal/Bez_Kick_33.pth):
`python -m bez_isaacgym_amd.train task=bez_kick num_envs=4096 headless=True max_iterations=6156 seed=S <overrides>` for every
(tag, overrides) x seed, the logged mean episode reward every 50 epochs as one CSV column each + a summary line per run.

    python tools/learning_curve.py --out profiles/r03_learning_curve.csv --seeds 42 43 44 45 --run default: --run ballcn155:task.sim.bez.ball_cn=155
"""
import argparse
import csv
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = re.compile(r"^epoch (\d+) frames (\d+) fps total (\d+).* mean_reward (-?[\d.]+)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/learning_curve.csv")
    ap.add_argument("--seeds", type=int, nargs="+", default=[42, 43, 44, 45])
    ap.add_argument("--epochs", type=int, default=6156)
    ap.add_argument("--every", type=int, default=50)
    ap.add_argument("--run", action="append", default=[], help="tag:override override ... (space separated after the colon)")
    a = ap.parse_args()
    cols, frames = {}, {}
    for spec in (a.run or ["default:"]):
        tag, _, ov = spec.partition(":")
        for seed in a.seeds:
            t0 = time.time()
            cmd = [sys.executable, "-m", "bez_isaacgym_amd.train", "task=bez_kick", "num_envs=4096", "headless=True", "max_iterations=%d" % a.epochs,
                   "seed=%d" % seed, "train.params.config.save_frequency=0"] + ov.split()
            p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True)
            col, fps = {}, 0
            for ln in p.stdout.splitlines():
                m = LINE.match(ln)
                if m:
                    e = int(m.group(1)); frames[e] = int(m.group(2)); fps = int(m.group(3)); col[e] = float(m.group(4))
            name = "%s_seed%d" % (tag, seed)
            cols[name] = col
            tail = [v for e, v in sorted(col.items())][-40:]
            print("%-28s epochs %d  last-2000-epoch mean %.2f  max %.2f  fps %d  (%.0f s, rc %d)" % (
                name, max(col) if col else 0, sum(tail) / max(len(tail), 1), max(col.values()) if col else float("nan"), fps, time.time() - t0, p.returncode), flush=True)
            if p.returncode:
                print(p.stderr[-1500:])
    epochs = sorted(e for e in frames if e % a.every == 0)
    with open(a.out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["epoch", "frames"] + ["mean_reward_" + n for n in cols])
        for e in epochs:
            w.writerow([e, frames[e]] + ["%.3f" % cols[n][e] if e in cols[n] else "" for n in cols])


if __name__ == "__main__":
    main()
