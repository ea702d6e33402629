#!/bin/bash
# Kernel stats of the PPO leg with and without --randomize on the same box: where the domain randomisation costs time.
set -e
export TMPDIR=/tmp
for mode in plain randomize; do
  rm -rf gpurun_out/rab_$mode
  flag=""; [ $mode = randomize ] && flag="--randomize"
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rab_$mode -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --ppo-epochs 10 $flag > gpurun_out/rab_$mode.log 2>&1
  echo "== $mode"
  python3 - $mode <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/rab_%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f ms" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print("%-92s calls %6s avg %8.2f us total %7.2f ms" % (r["Name"][:92], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
find gpurun_out -name "*agent_info.csv" -delete; find gpurun_out -path "*rab_*" -name "*kernel_trace.csv" -delete
