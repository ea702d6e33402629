#!/usr/bin/env python3
"""Idle time between consecutive kernels of the replayed PPO epoch, from a rocprofv3 kernel trace (what
would ONE launch per rollout step save?).  For every ordered pair of kernel names (a -> b) that occurs at least `--min` times: the median
gap between a's end and b's start, and the medians of the two durations.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --ppo-epochs 8
    python3 tools/graph_gaps.py gpurun_out/trace > profiles/r05_graph_gaps.txt
"""
import csv
import glob
import statistics
import sys
from collections import defaultdict


def short(n):
    for key, name in (("step_kernel_ws8", "sim step"), ("policy_forward_kernel<1", "rollout policy"), ("policy_forward_kernel<2", "train forward"),
                      ("policy_forward_kernel<0", "policy forward (values)"), ("policy_backward_kernel", "loss+backward"), ("wgrad_kernel", "wgrad"),
                      ("grad_reduce_all_kernel", "grad reduce"), ("adam_fused_kernel", "adam"), ("grad_norm_parts_kernel", "norm parts"),
                      ("gae_kernel", "gae"), ("prep_", "dataset prep"), ("ppo_rollout_post", "rollout post")):
        if key in n:
            return name
    return n.split("(")[0][-40:]


def main():
    root = sys.argv[1]
    min_count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    files = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    gaps, da, db = defaultdict(list), defaultdict(list), defaultdict(list)
    for (s0, e0, a), (s1, e1, b) in zip(rows, rows[1:]):
        if s1 - e0 < 200000:      # (an epoch boundary with its host sync is not a launch gap)
            gaps[(a, b)].append(s1 - e0); da[(a, b)].append(e0 - s0); db[(a, b)].append(e1 - s1)
    print("%-26s -> %-26s %7s %10s %10s %10s" % ("kernel a", "kernel b", "count", "gap us", "a us", "b us"))
    for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
        if len(v) >= min_count:
            print("%-26s -> %-26s %7d %10.2f %10.2f %10.2f" % (k[0], k[1], len(v), statistics.median(v) / 1e3, statistics.median(da[k]) / 1e3, statistics.median(db[k]) / 1e3))


if __name__ == "__main__":
    main()
