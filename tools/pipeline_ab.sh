# A/B on one GPU: PPO epochs pipelined (the report of epoch k read while epoch k + 1 is queued; default) or one at a time (BEZ_PPO_PIPELINE_EPOCHS=0)
set -e
for r in 1 2 3; do for v in 0 1; do
  BEZ_PPO_PIPELINE_EPOCHS=$v python3 bench.py --no-cpu-baseline --no-dp-path --steps 100 --warmup 10 --ppo-epochs 30 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read())['ppo']; print('pipeline_epochs=$v', round(d['value']), '%.3f ms' % d['epoch_ms'], 'rollout share %.3f' % d['rollout_share'])"
done; done
