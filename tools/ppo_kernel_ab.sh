#!/bin/bash
# Per-kernel durations of the PPO leg for library builds (tools/ab_build.sh) on the SAME box, from rocprofv3 kernel stats.
# (--no-dp-path: the data-parallel leg runs in a child process that rocprofv3 would profile into a second stats file of the same directory)
# usage: bash tools/ppo_kernel_ab.sh a b ...
set -e
export TMPDIR=/tmp
for n in "$@"; do
  export BEZ_SIM_LIB=$PWD/build_ab/$n.so   # (selected by environment: the in-tree library is never overwritten)
  rm -rf gpurun_out/kab_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kab_$n -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-dp-path --ppo-epochs ${PPO_AB_EPOCHS:-10} > gpurun_out/kab_$n.log 2>&1
  echo "== $n"
  python3 - "$n" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/kab_%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    name = r["Name"]
    if any(k in name for k in ("policy_forward", "policy_backward", "wgrad", "ppo_loss", "adam_fused", "grad_reduce", "step_kernel")):
        print("%-90s calls %6s avg %9.2f us" % (name[:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
find gpurun_out -name "*agent_info.csv" -delete; find gpurun_out -name "*kernel_trace.csv" -path "*kab_*" -delete
