#!/bin/bash
# timing experiments on the two-frame LDPC kernel: rebuild with the given macro sets (results wrong by design) and print per-layer cycles
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for X in "$@"; do
  touch $ROOT/sdr_receiver_dvb_t2_amd/csrc/ldpc_kernel2.hip
  make -C $ROOT/sdr_receiver_dvb_t2_amd/csrc -j8 EXTRA="$X" > /dev/null 2>&1
  echo "== EXTRA=$X"
  python $ROOT/tools/ldpc_phase_profile.py 1024 32 noise 2>&1 | grep -E "launch|cycles per" | cut -c1-330
done
