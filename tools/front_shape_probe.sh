#!/bin/bash
# front_farrow_decimate_kernel: "outputs-per-workgroup wavefronts-per-SIMD" pairs against duration (rebuilds the object on the box)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for CFG in "$@"; do
  set -- $CFG
  touch $ROOT/sdr_receiver_dvb_t2_amd/csrc/front_kernels.hip
  make -C $ROOT/sdr_receiver_dvb_t2_amd/csrc -j8 EXTRA="-DT2_FD_OUT=$1 -DT2_FD_WAVES=$2" > /dev/null 2>&1
  echo "== outputs per workgroup $1, wavefronts per SIMD asked $2"
  (cd $ROOT && python -m pytest tests/test_front_gpu.py -x -q 2>&1 | tail -1)
  bash $ROOT/tools/front_prof.sh 2>&1 | grep "front_farrow_decimate" | head -1
done
