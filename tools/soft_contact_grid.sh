#!/bin/bash
# Where in the soft / restitution-0 contact family is the high-reward kick learned reliably?  1500-epoch trainings, mean of the last 20 logged epochs.
# usage (GPU box): bash tools/soft_contact_grid.sh [outfile]
OUT=${1:-gpurun_out/r03_soft_contact_grid.txt}
: > $OUT
B=task.sim.bez
run() {
  name=$1; shift
  r=$(timeout -k 10 400 python -m bez_isaacgym_amd.train task=bez_kick num_envs=4096 headless=True max_iterations=1500 "$@" 2>&1 | grep "^epoch" | tail -20 | awk '{s+=$NF; n++} END {if (n) printf "%.2f", s/n; else printf "nan"}')
  echo "$name $r" | tee -a $OUT
  rm -rf runs
}
for seed in 61 62 63 64; do
run kn2500_ballcn55_s$seed seed=$seed $B.contact_kn=2500 $B.ball_cn=55
run kn2500_ballcn55_ct3000_s$seed seed=$seed $B.contact_kn=2500 $B.ball_cn=55 $B.contact_ct=3000
run kn5000_ballcn77_ct3000_s$seed seed=$seed $B.contact_kn=5000 $B.ball_cn=77 $B.contact_ct=3000
run kn5000_ballcn77_cn10_s$seed seed=$seed $B.contact_kn=5000 $B.ball_cn=77 $B.contact_cn=10
done
