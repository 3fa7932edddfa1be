#!/bin/bash
# one gpurun call: build, GPU test suite, bench with per-kernel rocprof stats, A/B of the equaliser's workgroup order
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8
timeout 600 python bench.py --steps 3 --no-cpu-baseline > gpurun_out/b3.json 2> gpurun_out/b3.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/b3.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"])
print(d.get("clamped_llr_variant"))
print([(k["stage"][:8], k["ms"], k.get("frac")) for k in d["roofline"]["kernels"]])
PY
tail -3 gpurun_out/b3.err
T2GPU_EQ_ROW_MAJOR=0 timeout 600 python bench.py --steps 3 --no-cpu-baseline --no-extra-legs > gpurun_out/b3_symorder.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/b3_symorder.json').read().strip().splitlines()[-1]); print('symbol order:', [(k['stage'][:8], k['ms']) for k in d['roofline']['kernels']])"
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_b3; timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_b3 -o b3 -- python $R/bench.py --no-cpu-baseline --no-extra-legs --steps 4 --warmup 1 > $R/gpurun_out/b3_rocprof.json 2> $R/gpurun_out/prof_b3.err
cd $R; python tools/rocprof_summary.py $(find gpurun_out/prof_b3 -name "*.db" | head -1) gpurun_out/b3_kernel_stats.txt | head -30
