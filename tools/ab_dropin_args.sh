#!/bin/bash
# same-box A/B of the slot-shaped path on argument sets of examples/t2gpu_rx_file (T2GPU_DROPIN_ARGS), alternating:
#   tools/ab_dropin_args.sh "--chain-one 1" "--chain-one 0"            (add --saturate as $SAT=1)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for r in 1 2 3; do
  for a in "$@"; do
    echo "[$a]: $(T2GPU_DROPIN_ARGS="$a" timeout 300 python bench.py --only-drop-in ${SAT:+--saturate} 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d.get("seconds"), d.get("ts_matches_sent"))')"
  done
done
