#!/bin/bash
# Sensitivity of the TRAINED reward to this build's free contact parameters: short trainings (1500 epochs = 197 M frames, the
# reward plateau is reached by ~1000) with one parameter changed at a time.  usage (GPU box): bash tools/train_sweep.sh
OUT=gpurun_out/r02_train_sweep.txt
: > $OUT
run() {
  name=$1; shift
  r=$(timeout -k 10 200 python -m bez_isaacgym_amd.train task=bez_kick num_envs=4096 headless=True max_iterations=1500 seed=42 "$@" 2>&1 | grep "^epoch" | tail -20 | awk '{s+=$NF; n++} END {printf "%.2f", s/n}')
  echo "$name $r" | tee -a $OUT
  rm -rf runs
}
run default
run kn_x4 task.sim.bez.contact_kn=80000
run kn_d4 task.sim.bez.contact_kn=5000
run cn_x4 task.sim.bez.contact_cn=80
run cn_d4 task.sim.bez.contact_cn=5
run ct_x10 task.sim.bez.contact_ct=10000
run ct_d10 task.sim.bez.contact_ct=100
run balldamp_005 task.sim.bez.ball_ang_damping=0.05
run veps_x10 task.sim.bez.contact_veps=0.1
run selfkn_x4 task.sim.bez.self_kn=12000
# second pass around the soft end (profiles/r02_train_sweep.txt holds both): contact_kn 2500 ... 10000, seeds, cn / ct at kn 5000
run kn2500 task.sim.bez.contact_kn=2500
run kn3500 task.sim.bez.contact_kn=3500
run kn5000_seed43 task.sim.bez.contact_kn=5000 seed=43
run kn7000 task.sim.bez.contact_kn=7000
run kn10000 task.sim.bez.contact_kn=10000
run kn5000_cn10 task.sim.bez.contact_kn=5000 task.sim.bez.contact_cn=10
run kn5000_cn40 task.sim.bez.contact_kn=5000 task.sim.bez.contact_cn=40
run kn5000_ct300 task.sim.bez.contact_kn=5000 task.sim.bez.contact_ct=300
run kn5000_ct3000 task.sim.bez.contact_kn=5000 task.sim.bez.contact_ct=3000
run kn5000_selfkn1000 task.sim.bez.contact_kn=5000 task.sim.bez.self_kn=1000
