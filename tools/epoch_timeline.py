#!/usr/bin/env python3
"""One replayed PPO epoch as the GPU saw it: every kernel that is not one of the seven big ones, with its start offset, duration and the idle time
in front of it, plus the epoch's span and total idle time.  Input: a rocprofv3 --kernel-trace CSV of `bench.py --no-dp-path` (the PPO leg).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-dp-path --ppo-epochs 12
    python3 tools/epoch_timeline.py gpurun_out/tl
"""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
BIG = (("step_kernel_ws8", "sim"), ("policy_forward_kernel<1", "policy"), ("policy_forward_kernel<2", "fwd"), ("policy_backward", "bwd"), ("wgrad_kernel", "wgrad"),
       ("grad_reduce_all", "reduce"), ("adam_fused", "adam"))


def short(n):
    for k, v in BIG:
        if k in n:
            return v
    if "policy_forward_kernel<0" in n:
        return "fwd0 (last values)"
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")
    if "FillFunctor" in n:
        return "fill<%s>" % n.split("FillFunctor<")[1].split(">")[0]
    return n.split("(")[0][:60]


idx = [i for i, r in enumerate(rows) if "policy_forward_kernel<0" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
prev_end, idle, big = None, 0.0, {}
print("%10s  %-62s %8s %8s" % ("start us", "kernel", "us", "idle us"))
for r in seg:
    s, e, n = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])
    gap = max(0.0, (s - prev_end) / 1e3) if prev_end else 0.0
    idle += gap
    if n in dict(BIG).values():
        c = big.setdefault(n, [0, 0.0]); c[0] += 1; c[1] += (e - s) / 1e3
        if gap > 1.0:
            print("%10.1f  %-62s %8.2f %8.2f" % ((s - t0) / 1e3, n, (e - s) / 1e3, gap))
    else:
        print("%10.1f  %-62s %8.2f %8.2f   grid %s" % ((s - t0) / 1e3, n, (e - s) / 1e3, gap, r["Grid_Size_X"]))
    prev_end = max(prev_end or 0, e)
print("epoch span %.1f us, %d kernels, idle %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, len(seg), idle))
for n, (c, t) in big.items():
    print("   %-8s x %3d  %8.1f us" % (n, c, t))
