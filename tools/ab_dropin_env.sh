#!/bin/bash
# same-box A/B of the slot-shaped path on environment settings, alternating: tools/ab_dropin_env.sh "X=1" "T2GPU_LDPC_UNCACHED=1" ...
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
run() { echo "[$1]: $(env $1 timeout 300 python bench.py --only-drop-in ${SAT:+--saturate} 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d.get("seconds"))')"; }
for r in 1 2 3; do for a in "$@"; do run "$a"; done; done
