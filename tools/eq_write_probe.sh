#!/bin/bash
# WRITE_SIZE / duration of eq_data_kernel against the number of resident workgroups per CU (T2GPU_EQ_LDS_PAD), config-2 bench.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for PAD in ${@:-0 30000 60000}; do
  rm -rf $ROOT/gpurun_out/eqw_$PAD
  T2GPU_EQ_LDS_PAD=$PAD rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/eqw_$PAD -o p -- python $ROOT/bench.py --config 2 --no-cpu-baseline --no-extra-legs --steps 2 --warmup 1 > /dev/null 2>&1
  python - <<PY
import csv,glob
rows=[r for f in glob.glob("$ROOT/gpurun_out/eqw_$PAD/**/*counter_collection.csv",recursive=True) for r in csv.DictReader(open(f))]
eq=[r for r in rows if "eq_data_kernel" in r["Kernel_Name"] and r["Counter_Name"]=="WRITE_SIZE"]
big=sorted(float(r["Counter_Value"]) for r in eq)[len(eq)//2:]
tr=[r for f in glob.glob("$ROOT/gpurun_out/eqw_$PAD/**/*kernel_trace.csv",recursive=True) for r in csv.DictReader(open(f)) if "eq_data_kernel" in r["Kernel_Name"]]
d=sorted((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in tr)[len(tr)//2:]
print("pad $PAD: data-symbol launches WRITE_SIZE avg %.0f KiB, duration avg %.1f us (%d launches)"%(sum(big)/len(big), sum(d)/len(d), len(d)))
PY
done
