#!/bin/bash
# eq_data_kernel: segments per workgroup / lanes / workgroups per CU against duration and WRITE_SIZE (rebuilds the two objects on the box)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for CFG in "$@"; do
  set -- $CFG
  touch $ROOT/sdr_receiver_dvb_t2_amd/csrc/ofdm_kernels.hip $ROOT/sdr_receiver_dvb_t2_amd/csrc/t2gpu_ofdm.cpp
  make -C $ROOT/sdr_receiver_dvb_t2_amd/csrc -j8 EXTRA="-DT2_EQ_GROUP=$1 -DT2_EQ_THREADS=$2 -DT2_EQ_WGS_PER_CU=$3" > /dev/null 2>&1
  echo "== group $1 threads $2 workgroups/CU $3"
  bash $ROOT/tools/eq_write_probe.sh 0 2>&1 | tail -1
done
