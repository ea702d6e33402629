#!/bin/bash
# SQ / LDS counters of the PPO leg's kernels (per launch averages), one rocprofv3 --pmc pass per counter group.
# usage: bash tools/ppo_pmc.sh <out-tag>
set -e
export TMPDIR=/tmp
T=${1:-pmc}
B="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-dp-path --ppo-epochs 4"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/${T}_g$i -- $B > gpurun_out/${T}_g$i.log 2>&1 || echo "group $i failed: $grp"
done
python3 - "$T" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/%s_g*/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(s in k for s in ("policy_forward", "policy_backward", "wgrad_kernel", "grad_reduce_all", "adam_fused")):
            continue
        k = k[:70]
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v, n = acc[k][c]
        print("   %-32s %14.0f  (%d launches)" % (c, v / n, n))
PY
find gpurun_out -name "*agent_info.csv" -delete
