#!/usr/bin/env python3
"""Idle time between consecutive kernels of the PPO leg: python tools/gap_probe.py <kernel_trace.csv> (rocprofv3 --kernel-trace)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last epochs only (steady state, graph replays): take the final 40 % of dispatches
rows = rows[int(len(rows) * 0.6):]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
gaps = collections.defaultdict(lambda: [0, 0])
tot_gap = 0
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g > 0:
        tot_gap += g
    k = a["Kernel_Name"][:50] + " -> " + b["Kernel_Name"][:50]
    gaps[k][0] += max(g, 0); gaps[k][1] += 1
print("dispatches %d  span %.2f ms  busy %.2f ms (%.0f %%)  idle between kernels %.2f ms" % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, tot_gap / 1e6))
for k, (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print("%8.1f us total %6d x  %5.2f us  %s" % (g / 1e3, n, g / 1e3 / n, k))
