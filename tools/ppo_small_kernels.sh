#!/bin/bash
# The small launches of a PPO epoch (everything but the six big kernels), per epoch, from rocprofv3 kernel stats of the PPO leg.
set -e
export TMPDIR=/tmp
rm -rf gpurun_out/small
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/small -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-dp-path --ppo-epochs 10 > gpurun_out/small.log 2>&1
export SMALL_EPOCHS=$(python3 -c "import json; d=json.loads(open('gpurun_out/small.log').read().strip().splitlines()[-1]); print(d['ppo']['epochs'] + 4)")
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/small/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
import os
ep = float(os.environ.get('SMALL_EPOCHS', 14))   # timed epochs (from the bench line) + 4 warm-up / capture epochs
tot = 0
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    n = r["Name"]
    if any(k in n for k in ("step_kernel_ws8", "wgrad_kernel", "policy_", "ppo_loss", "grad_reduce_all", "adam_fused")): continue
    per_epoch = float(r["TotalDurationNs"]) / 1e3 / ep
    tot += per_epoch
    if per_epoch > 2.0:
        print("%-100s calls/epoch %5.1f avg %7.2f us  per epoch %6.1f us" % (n[:100], float(r["Calls"]) / ep, float(r["AverageNs"]) / 1e3, per_epoch))
print("sum of the small kernels per epoch: %.0f us" % tot)
PY
find gpurun_out -name "*agent_info.csv" -delete; find gpurun_out/small -name "*kernel_trace.csv" -delete
