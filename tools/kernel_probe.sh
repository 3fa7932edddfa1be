#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / duration of the kernels matching $1 over a short bench run (two rocprofv3 --pmc passes)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PAT=${1:-ti_block}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $ROOT/gpurun_out/kp_$C
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $ROOT/gpurun_out/kp_$C -o p -- python $ROOT/bench.py --no-cpu-baseline --no-extra-legs --steps 2 --warmup 1 > /dev/null 2>&1
  python - <<PY
import csv,glob
rows=[r for f in glob.glob("$ROOT/gpurun_out/kp_$C/**/*counter_collection.csv",recursive=True) for r in csv.DictReader(open(f))]
v=[float(r["Counter_Value"]) for r in rows if "$PAT" in r["Kernel_Name"] and r["Counter_Name"]=="$C"]
tr=[r for f in glob.glob("$ROOT/gpurun_out/kp_$C/**/*kernel_trace.csv",recursive=True) for r in csv.DictReader(open(f)) if "$PAT" in r["Kernel_Name"]]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in tr]
print("$PAT $C avg %.0f KiB over %d launches, duration avg %.1f us"%(sum(v)/max(len(v),1), len(v), sum(d)/max(len(d),1)))
PY
done
