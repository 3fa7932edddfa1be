#!/bin/bash
# same-box comparison of builds of the library on the slot-shaped path: sdr_receiver_dvb_t2_amd/libt2gpu.so ("new") against every
# libt2gpu_*.so.keep beside it, alternating, bench.py --only-drop-in (the example program finds the library through its rpath)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT/sdr_receiver_dvb_t2_amd
cp libt2gpu.so /tmp/ab_new.so
for r in 1 2 3; do
  for f in /tmp/ab_new.so libt2gpu_*.so.keep; do
    cp $f libt2gpu.so.tmp && mv libt2gpu.so.tmp libt2gpu.so
    echo "$(basename $f): $(cd $ROOT && timeout 300 python bench.py --only-drop-in $1 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d.get("seconds"))')"
  done
done
cp /tmp/ab_new.so libt2gpu.so
