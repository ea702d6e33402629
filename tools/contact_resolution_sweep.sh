#!/bin/bash
# Is the trained reward level decided by how well the substep RESOLVES the ground contact (omega h <= 1), not by a ball catapult?
# 1500-epoch trainings, mean reward of the last 20 logged epochs.   usage (GPU box): bash tools/contact_resolution_sweep.sh [outfile] [seeds...]
OUT=${1:-gpurun_out/r03_contact_resolution.txt}
shift
SEEDS=${@:-42 43}
: > $OUT
run() {
  name=$1; shift
  r=$(timeout -k 10 400 python -m bez_isaacgym_amd.train task=bez_kick num_envs=4096 headless=True max_iterations=1500 "$@" 2>&1 | grep "^epoch" | tail -20 | awk '{s+=$NF; n++} END {if (n) printf "%.2f", s/n; else printf "nan"}')
  echo "$name $r" | tee -a $OUT
  rm -rf runs
}
B=task.sim.bez
for seed in $SEEDS; do
run default_s$seed seed=$seed
run ground5000_ballstiff_zeta1_s$seed seed=$seed $B.contact_kn=5000 $B.ball_kn=20000 $B.ball_cn=155
run all5000_ballzeta1_s$seed seed=$seed $B.contact_kn=5000 $B.ball_cn=77
run all5000_s$seed seed=$seed $B.contact_kn=5000
run substeps4_s$seed seed=$seed task.sim.substeps=4
run substeps4_ballzeta1_s$seed seed=$seed task.sim.substeps=4 $B.ball_cn=155
done
