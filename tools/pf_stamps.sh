# stamps of the forward-only and rollout forms of the policy launch, 64- and 32-row tiles (diagnostic build; tools/policy_stamp_probe.py)
set -e
for v in ${PF_STAMP_ROWS:-64 32}; do for r in "" 1 2; do
  echo "=== rows $v ROLL=${r:-0}"
  BEZ_PF_ROWS=$v PACKED=1 ROLL=$r timeout -k 10 300 python3 tools/policy_stamp_probe.py
done; done
