#!/bin/bash
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -5
timeout 600 python bench.py --steps 3 --no-cpu-baseline > gpurun_out/b4.json 2> gpurun_out/b4.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/b4.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"].get("valu_issue_frac"))
print({k: v for k, v in d["clamped_llr_variant"].items() if k != "host_end_counters"})
PY
tail -3 gpurun_out/b4.err
for c in 2 4 5; do timeout 600 python bench.py --config $c --steps 3 --no-cpu-baseline --no-extra-legs > gpurun_out/bench_config${c}_r03.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/bench_config${c}_r03.json').read().strip().splitlines()[-1]); print($c, d['value'], d['ms_per_step'], [(k['stage'][:6], k['ms']) for k in d['roofline']['kernels']])"; done
