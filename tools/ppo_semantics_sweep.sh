#!/bin/bash
# rl_games 1.1.3 update-loop semantics (schedule_type legacy|standard x update_mu_sigma): short trainings, mean of the last 20 logged epochs.
# usage (GPU box): bash tools/ppo_semantics_sweep.sh [outfile] [epochs] [seeds...]
OUT=${1:-gpurun_out/r03_ppo_semantics.txt}
EP=${2:-1500}
shift; shift
SEEDS=${@:-42 43}
: > $OUT
run() {
  name=$1; shift
  r=$(timeout -k 10 400 python -m bez_isaacgym_amd.train task=bez_kick num_envs=4096 headless=True max_iterations=$EP "$@" 2>&1 | grep "^epoch" | tail -20 | awk '{s+=$NF; n++} END {if (n) printf "%.2f", s/n; else printf "nan"}')
  echo "$name $r" | tee -a $OUT
  rm -rf runs
}
for seed in $SEEDS; do
run legacy_update_s$seed seed=$seed
run legacy_noupdate_s$seed seed=$seed train.params.config.update_mu_sigma=False
run standard_update_s$seed seed=$seed train.params.config.schedule_type=standard
run standard_noupdate_s$seed seed=$seed train.params.config.schedule_type=standard train.params.config.update_mu_sigma=False
done
