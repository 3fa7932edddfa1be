#!/bin/bash
# per-launch duration, grid and FETCH_SIZE / WRITE_SIZE of the equaliser kernels over a short bench run, for each T2GPU_EQ_SPLITS in "$@"
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for NS in "$@"; do
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $ROOT/gpurun_out/kp_$C
  T2GPU_EQ_SPLITS=$NS rocprofv3 --kernel-trace --pmc $C --output-format csv -d $ROOT/gpurun_out/kp_$C -o p -- python $ROOT/bench.py --no-cpu-baseline --no-extra-legs --steps 2 --warmup 1 > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
rows=[r for f in glob.glob("$ROOT/gpurun_out/kp_$C/**/*counter_collection.csv",recursive=True) for r in csv.DictReader(open(f))]
tr=[r for f in glob.glob("$ROOT/gpurun_out/kp_$C/**/*kernel_trace.csv",recursive=True) for r in csv.DictReader(open(f)) if "eq_" in r["Kernel_Name"]]
g=collections.defaultdict(list)
for r in tr: g[(r["Kernel_Name"][:40], r.get("Grid_Size_X", r.get("Grid_Size","?")))].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
c=collections.defaultdict(list)
for r in rows:
    if "eq_" in r["Kernel_Name"] and r["Counter_Name"]=="$C": c[(r["Kernel_Name"][:40], r.get("Grid_Size_X", r.get("Grid_Size","?")))].append(float(r["Counter_Value"]))
for k in sorted(g): print("splits=$NS", k, "launches %d avg %.1f us"%(len(g[k]), sum(g[k])/len(g[k])), "$C avg %.0f KiB"%(sum(c[k])/max(len(c[k]),1)))
PY
done
done
