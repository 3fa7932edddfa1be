#!/bin/bash
# same-box comparison of builds of the library on the slot-shaped path: sdr_receiver_dvb_t2_amd/libt2gpu.so ("new") against every
# libt2gpu_*.so.keep beside it, alternating, bench.py --only-drop-in. The example program finds the library under test through
# LD_LIBRARY_PATH (which goes before its run path); bench.py itself keeps the tree's library (it only synthesises the frames).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT/sdr_receiver_dvb_t2_amd
for r in 1 2 3; do
  for f in libt2gpu.so libt2gpu_*.so.keep; do
    rm -rf /tmp/ab_dir && mkdir -p /tmp/ab_dir && cp $f /tmp/ab_dir/libt2gpu.so
    echo "$(basename $f): $(cd $ROOT && LD_LIBRARY_PATH=/tmp/ab_dir timeout 300 python bench.py --only-drop-in $1 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d.get("seconds"))')"
  done
done
