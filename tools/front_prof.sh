#!/bin/bash
# rocprofv3 kernel rows of the sample-rate front end over a short bench run
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
rm -rf $R/gpurun_out/prof_front
(cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_front -o fr -- python $R/bench.py --no-cpu-baseline --no-extra-legs --steps 4 --warmup 1 > $R/gpurun_out/front_bench.json 2> /dev/null)
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_front -name "*.db" | head -1) $R/gpurun_out/front_stats.txt > /dev/null
grep -E "front_|eq_data|fft_fwd|ti_block|demap|p1_|cp_corr" $R/gpurun_out/front_stats.txt | cut -c1-60,75-130
