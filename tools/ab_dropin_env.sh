#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "[$1]: $(env $1 timeout 300 python bench.py --only-drop-in 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d.get("seconds"))')"; }
for r in 1 2; do
  run "X=1"
  run "HIP_FORCE_DEV_KERNARG=1"
  run "GPU_MAX_HW_QUEUES=8"
  run "HIP_FORCE_DEV_KERNARG=1 GPU_MAX_HW_QUEUES=8"
  run "HSA_ENABLE_INTERRUPT=0"
  run "ROC_ACTIVE_WAIT_TIMEOUT=1000"
done
