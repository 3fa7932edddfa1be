#!/bin/bash
# one gpurun call: build, ubench, GPU test suite, bench (+ cooperative-launch A/B). Output under gpurun_out/.
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
tools/ubench/valu_rate > gpurun_out/valu_rate.txt 2>&1; grep -v "^json" gpurun_out/valu_rate.txt | head -40
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8
timeout 600 python bench.py --steps 3 --no-cpu-baseline > gpurun_out/b2.json 2> gpurun_out/b2.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/b2.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"])
print(d.get("clamped_llr_variant"))
print([(k["stage"][:8], k["ms"]) for k in d["roofline"]["kernels"]])
PY
tail -3 gpurun_out/b2.err
T2GPU_LDPC_COOPERATIVE=0 timeout 600 python bench.py --steps 3 --no-cpu-baseline --no-extra-legs > gpurun_out/b2_plain.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/b2_plain.json').read().strip().splitlines()[-1]); print('plain launch:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
