#!/bin/bash
# SQ counters of the LDPC kernel alone (tools/ldpc_time.py: 7680 all-noise frames, 25 sweeps), packed and one-frame variants.
#   bash tools/ldpc_pmc.sh  -> gpurun_out/ldpc_pmc_{1,0}.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  T2GPU_LDPC_PACKED=$v rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $ROOT/gpurun_out/ldpc_pmc_$v -o p -- python $ROOT/tools/ldpc_time.py > $ROOT/gpurun_out/ldpc_pmc_$v.log 2>&1
  python - <<PY > $ROOT/gpurun_out/ldpc_pmc_$v.txt
import csv, glob, collections
rows = []
for f in glob.glob("$ROOT/gpurun_out/ldpc_pmc_$v/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k in acc:
    if "ldpc" not in k: continue
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-22s %.4e per launch (%d launches)" % (c, v / n[(k, c)], n[(k, c)]))
PY
  cat $ROOT/gpurun_out/ldpc_pmc_$v.txt
done
