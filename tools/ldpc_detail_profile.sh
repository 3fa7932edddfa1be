#!/bin/bash
# diagnostics: rebuild the LDPC kernel with T2_PROF_DETAIL=$1 (1 = PAIR layers, 2 = GENERIC layers split into phase A / serial / finish;
# 3 = PLAIN layers split into load / compute+store / barrier)
# and print the raw profiler slots; restores the production build afterwards.
set -e
cd "$(dirname "$0")/.."
K=sdr_receiver_dvb_t2_amd/csrc
touch $K/ldpc_kernel.hip && make -s -C $K EXTRA=-DT2_PROF_DETAIL=$1 >/dev/null
python tools/ldpc_phase_profile.py 2048 32 | tail -8
T2GPU_LDPC_BLOCKS_PER_CU=1 python tools/ldpc_phase_profile.py 2048 32 | tail -8
touch $K/ldpc_kernel.hip && make -s -C $K >/dev/null
