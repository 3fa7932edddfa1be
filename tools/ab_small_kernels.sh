#!/bin/bash
# same-box A/B of the small kernels: builds the library twice (default, then with $1 as EXTRA) and prints their rocprofv3 rows
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for v in new old; do
  if [ $v = old ]; then (cd sdr_receiver_dvb_t2_amd/csrc && touch fec_kernels.hip ofdm_kernels.hip front_kernels.hip p1_kernels.hip && make -j8 EXTRA="$1" > /dev/null 2>&1); fi
  rm -rf gpurun_out/prof_ab_$v
  (cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ab_$v -o ab -- python $R/bench.py --no-cpu-baseline --no-clamped-variant --steps 6 --warmup 1 > $R/gpurun_out/ab_$v.json 2> /dev/null)
  python tools/rocprof_summary.py $(find gpurun_out/prof_ab_$v -name "*.db" | head -1) gpurun_out/ab_${v}_stats.txt > /dev/null
  echo "== $v"; sed -n 4,24p gpurun_out/ab_${v}_stats.txt | cut -c1-40,75-130
done
