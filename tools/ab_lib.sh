#!/bin/bash
# same-box A/B of two builds of the library: libt2gpu.so (new) against libt2gpu_old.so.keep, alternating, LDPC noise load
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT/sdr_receiver_dvb_t2_amd
cp libt2gpu.so /tmp/new.so; cp libt2gpu_old.so.keep /tmp/old.so
for r in 1 2 3; do
  for v in old new; do
    cp /tmp/$v.so libt2gpu.so
    echo "$v: $(python $ROOT/tools/ldpc_phase_profile.py 7680 32 noise 2>&1 | grep launch)"
  done
done
cp /tmp/new.so libt2gpu.so
