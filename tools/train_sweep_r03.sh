#!/bin/bash
# Round 3: how the TRAINED reward depends on the ball's contact damping (PhysX restitution is 0; the round-2 defaults gave the
# ball <-> foot spring a damping ratio of 0.13).  Short trainings (1500 epochs = 197 M frames), mean of the last 20 logged epochs.
# usage (GPU box): bash tools/train_sweep_r03.sh [outfile]
OUT=${1:-gpurun_out/r03_train_sweep.txt}
: > $OUT
run() {
  name=$1; shift
  r=$(timeout -k 10 240 python -m bez_isaacgym_amd.train task=bez_kick num_envs=4096 headless=True max_iterations=1500 seed=42 "$@" 2>&1 | grep "^epoch" | tail -20 | awk '{s+=$NF; n++} END {if (n) printf "%.2f", s/n; else printf "nan"}')
  echo "$name $r" | tee -a $OUT
  rm -rf runs
}
run default
run ballcn80 task.sim.bez.ball_cn=80
run ballcn155 task.sim.bez.ball_cn=155
run ballcn155_s43 task.sim.bez.ball_cn=155 seed=43
run ballcn300 task.sim.bez.ball_cn=300
run ballcn600 task.sim.bez.ball_cn=600
run ballkn5000_cn77 task.sim.bez.ball_kn=5000 task.sim.bez.ball_cn=77
run ballkn5000_cn77_s43 task.sim.bez.ball_kn=5000 task.sim.bez.ball_cn=77 seed=43
run ballkn80000_cn310 task.sim.bez.ball_kn=80000 task.sim.bez.ball_cn=310
run ballcn155_allcn240 task.sim.bez.ball_cn=155 task.sim.bez.contact_cn=240
