# A/B on one GPU: the data-parallel PPO path with its update as S + 1 graph segments (default) or launched eagerly (BEZ_PPO_DP_EAGER_UPDATE=1)
set -e
for r in 1 2 3; do
for v in 0 1; do
  echo "== dp_eager_update=$v"
  BEZ_PPO_DP_EAGER_UPDATE=$v MASTER_ADDR=127.0.0.1 MASTER_PORT=2961$v RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 python3 bench.py --dp-path-child --ppo-epochs 30 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ('dp_path_epoch_ms','dp_path_samples_per_s','dp_path_rollout_share')})"
done; done
