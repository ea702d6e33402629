#!/bin/bash
# L2 hits / misses of the weight-gradient kernel with the workgroup ids in plan order (BEZ_WGRAD_XCD=0) and with the XCD-aware placement (default).
export TMPDIR=/tmp
B="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-dp-path --ppo-epochs 4"
for x in ${WGL2_SET:-0 4}; do
  BEZ_WGRAD_XCD=$x rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/wgl2_$x -- $B > gpurun_out/wgl2_$x.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
import os
for x in [int(v) for v in os.environ.get("WGL2_SET", "0 4").split()]:
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("gpurun_out/wgl2_%d/**/*counter_collection.csv" % x, recursive=True):
        for r in csv.DictReader(open(f)):
            if "wgrad_kernel" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    n = max(1, max(v[1] for v in acc.values()) if acc else 1)
    d = {k: v[0] / v[1] for k, v in acc.items()}
    h, m = d.get("TCC_HIT_sum", 0), d.get("TCC_MISS_sum", 0)
    print("BEZ_WGRAD_XCD=%d  wgrad_kernel per launch: L2 hits %.0f  misses %.0f  hit rate %.1f %%  (misses x 128 B = %.1f MB)  %s" % (x, h, m, 100 * h / max(h + m, 1), m * 128 / 1e6, {k: round(v) for k, v in d.items() if k.startswith("TCC_EA")}))
PY
