#!/bin/bash
# Round 6 (VERDICT r5 item 7 i): where the LDPC kernel's parked wave time goes, and where its per-link message traffic is served.
# Separate rocprofv3 --pmc passes (MI355X_MICROARCH.md "rocprofv3 PMC slots": 8 SQ slots, 4 TCC slots per pass), --kernel-trace only.
#   bash tools/pmc_ldpc_wait.sh <tag>   -> gpurun_out/pmcw_<tag>_*/ and gpurun_out/pmcw_<tag>.txt
set -e
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CMD="python $ROOT/bench.py --no-cpu-baseline --no-extra-legs --steps 2 --warmup 1"
cd /tmp && export TMPDIR=/tmp
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $ROOT/gpurun_out/pmcw_${TAG}_$1 -o p -- $CMD > $ROOT/gpurun_out/pmcw_${TAG}_$1.json 2> $ROOT/gpurun_out/pmcw_${TAG}_$1.err || true; }
run wait "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
run inst "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES SQ_WAVES"
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
run tcc2 "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_RDREQ_32B_sum"
cd $ROOT
python - <<PY > gpurun_out/pmcw_${TAG}.txt
import csv, glob, os
from collections import defaultdict
root = "$ROOT"
for name in ("wait", "inst", "tcc", "tcc2"):
    files = glob.glob(os.path.join(root, "gpurun_out", "pmcw_${TAG}_%s" % name, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print("# pass %s: no output" % name); continue
    per = defaultdict(lambda: defaultdict(float)); dur = {}
    for row in csv.DictReader(open(files[0])):
        if "ldpc_decode" not in row["Kernel_Name"]: continue
        key = int(row["Dispatch_Id"])
        per[key][row["Counter_Name"]] += float(row["Counter_Value"])
        dur[key] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
    ks = sorted(per)
    print("## pass %s: ldpc_decode2_kernel, %d launches, avg %.1f us" % (name, len(ks), sum(dur.values()) / max(1, len(ks))))
    for c in sorted({c for k in ks for c in per[k]}):
        print("%-28s %s" % (c, " ".join("%.6g" % per[k][c] for k in ks)))
PY
cat gpurun_out/pmcw_${TAG}.txt
