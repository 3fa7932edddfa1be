#!/bin/bash
# equaliser: output-range kernel vs segment-group kernel -- parity tests, then the stage time for each (ranges, lanes) shape
# usage: gpurun -- 'bash tools/eq_shapes_ab.sh 0:1024 2:1024 3:512'   (T2GPU_EQ_SPLITS:T2GPU_EQ_THREADS; 0 = segment-group kernel)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
timeout 900 python -m pytest tests/test_ofdm_gpu.py tests/test_chain_gpu.py tests/test_receiver_gpu.py -q -m gpu -x 2>&1 | tail -5
for cfg in ${@:-0:1024 2:1024 3:1024 3:512 4:512 5:512 6:512}; do
  set -- ${cfg/:/ }
  echo "== T2GPU_EQ_SPLITS=$1 T2GPU_EQ_THREADS=$2"
  T2GPU_EQ_SPLITS=$1 T2GPU_EQ_THREADS=$2 timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], [ (k['stage'],k['ms']) for k in d['roofline']['kernels'] if k['stage'] in ('equalise','ti','fft')])"
done
