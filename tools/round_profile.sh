#!/bin/bash
# One call on the GPU box: GPU test suite, bench line, rocprofv3 kernel stats and the three PMC passes, all under gpurun_out/ with tag $1.
# Afterwards copy gpurun_out/{bench_$1.json, ${1}_kernel_stats.txt, pmc_$1.txt, bench_under_rocprof_$1.json} into profiles/.
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
rm -rf gpurun_out/prof_$TAG; mkdir -p gpurun_out/prof_$TAG
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --steps 10 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o $TAG -- python $R/bench.py --no-cpu-baseline --no-extra-legs --steps 6 --warmup 1 \
    > $R/gpurun_out/bench_under_rocprof_$TAG.json 2> $R/gpurun_out/prof_$TAG.err
cd $R
python tools/rocprof_summary.py $(find gpurun_out/prof_$TAG -name "*.db" | head -1) gpurun_out/${TAG}_kernel_stats.txt | head -8
timeout 900 bash tools/pmc_passes.sh $TAG 2>&1 | sed -n 3,15p
grep -A1 "pass write: ldpc" gpurun_out/pmc_$TAG.txt
python -c "import json; d=json.loads(open('gpurun_out/bench_$TAG.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['share_of_step'], d['clamped_llr_variant']['msamples_per_s'], d['cpu_baseline']['value'])"
