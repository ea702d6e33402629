#!/usr/bin/env python3
"""Kernel sequence of one PPO minibatch step (between two adam_commit kernels) or of one rollout step (between two sim kernels):
python tools/seq_probe.py <kernel_trace.csv> [update|rollout]   (rocprofv3 --kernel-trace of `python3 bench.py ...`)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * 0.8):]
key = "adam_commit" if (len(sys.argv) > 2 and sys.argv[2] == "update") else "step_kernel_ws8"
idx = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
for a, b in zip(idx, idx[1:]):
    if b - a < 80:
        for r in rows[a + 1:b + 1]:
            print("%8.1f us  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:120]))
        break
