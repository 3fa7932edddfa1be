#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for cfg in ${@:-3:512:0}; do
  IFS=: read a b c <<< "$cfg"
  echo "== SPLITS=$a THREADS=$b DBG=$c"
  T2GPU_EQ_SPLITS=$a T2GPU_EQ_THREADS=$b T2GPU_EQ_DBG=$c timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], [ (k['stage'],k['ms']) for k in d['roofline']['kernels'] if k['stage'] in ('equalise',)])"
done
