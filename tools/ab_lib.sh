#!/bin/bash
# same-box comparison of builds of the library: sdr_receiver_dvb_t2_amd/libt2gpu.so ("new") against every libt2gpu_*.so.keep beside it,
# alternating, LDPC all-noise load of the bench (7680 frames, 25 sweeps); $1 = noise | decodable, $2 = code rate id of the 64800 code
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT/sdr_receiver_dvb_t2_amd
cp libt2gpu.so /tmp/ab_new.so
for r in 1 2 3; do
  for f in /tmp/ab_new.so libt2gpu_*.so.keep; do
    cp $f libt2gpu.so.tmp && mv libt2gpu.so.tmp libt2gpu.so
    echo "$(basename $f): $(timeout 120 python $ROOT/tools/ldpc_phase_profile.py 7680 32 ${1:-noise} ${2:-3} 2>&1 | grep launch)"
  done
done
cp /tmp/ab_new.so libt2gpu.so
