#!/bin/bash
# Distribution over seeds of the trained reward (2000 epochs = 262 M frames, mean of the last 20 logged epochs) for the default contact
# constants and for the resolved / restitution-0 variant (contact_kn 5000: omega h ~ 1 at the 8.3 ms substep; ball_cn 77: ball damping ratio 1).
# usage (GPU box): bash tools/seed_distribution.sh [outfile] [seeds...]
OUT=${1:-gpurun_out/r03_seed_distribution.txt}
shift
SEEDS=${@:-1 2 3 4 5 6 7 8}
: > $OUT
run() {
  name=$1; shift
  r=$(timeout -k 10 400 python -m bez_isaacgym_amd.train task=bez_kick num_envs=4096 headless=True max_iterations=2000 "$@" 2>&1 | grep "^epoch" | tail -20 | awk '{s+=$NF; n++} END {if (n) printf "%.2f", s/n; else printf "nan"}')
  echo "$name $r" | tee -a $OUT
  rm -rf runs
}
B=task.sim.bez
for seed in $SEEDS; do
run default_s$seed seed=$seed
run kn5000_ballcn77_s$seed seed=$seed $B.contact_kn=5000 $B.ball_cn=77
done
