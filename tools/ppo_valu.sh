#!/bin/bash
# Vector-instruction counts of the PPO leg's kernels (per launch and per wave): the MFMA policy kernels are bound by VALU issue.
set -e
export TMPDIR=/tmp
T=${1:-valu}
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_MFMA --output-format csv -d gpurun_out/${T} -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --ppo-epochs 4 > gpurun_out/${T}.log 2>&1
python3 - "$T" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % sys.argv[1], recursive=True):
    for r in csv.DictReader(open(f)):
        a = acc[r["Kernel_Name"][:80]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    c = {n: v[0] / v[1] for n, v in acc[k].items()}
    if c.get("SQ_WAVES", 0) > 0 and c.get("SQ_INSTS_VALU", 0) > 1e5:
        print("%-80s waves %6d  VALU/wave %7.0f  SALU/wave %6.0f  MFMA/wave %5.0f" % (k, c["SQ_WAVES"], c["SQ_INSTS_VALU"] / c["SQ_WAVES"], c.get("SQ_INSTS_SALU", 0) / c["SQ_WAVES"], c.get("SQ_INSTS_MFMA", 0) / c["SQ_WAVES"]))
PY
find gpurun_out -name "*agent_info.csv" -delete
