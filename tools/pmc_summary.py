#!/usr/bin/env python3
"""Summarise tools/pmc_passes.sh output: per-launch counter values of the LDPC kernel (and totals of the other t2gpu kernels)."""
import csv, glob, os, sys
from collections import defaultdict
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
print("# rocprofv3 --pmc passes (tools/pmc_passes.sh %s): bench.py --no-cpu-baseline --no-clamped-variant --steps 2 --warmup 1" % tag)
print("# FETCH_SIZE / WRITE_SIZE in KiB as reported; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts wide coalesced reads at 1/2")
for name in ("sq", "fetch", "write"):
    files = glob.glob(os.path.join(root, "gpurun_out", "pmc_%s_%s" % (tag, name), "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print("# pass %s: no output" % name)
        continue
    per = defaultdict(lambda: defaultdict(float))        # (kernel, dispatch) -> counter -> value
    dur = {}
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"]
        key = (k, int(row["Dispatch_Id"]))
        per[key][row["Counter_Name"]] += float(row["Counter_Value"])
        dur[key] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
    ldpc = sorted(k for k in per if "ldpc_decode_kernel" in k[0])
    print("## pass %s: ldpc_decode_kernel, one column per launch (%d launches), then duration in us" % (name, len(ldpc)))
    counters = sorted({c for k in ldpc for c in per[k]})
    for c in counters:
        print("%-24s %s" % (c, " ".join("%.6g" % per[k][c] for k in ldpc)))
    print("%-24s %s" % ("dur_us", " ".join("%.1f" % dur[k] for k in ldpc)))
    other = defaultdict(lambda: defaultdict(float))
    for (k, d), cs in per.items():
        if "ldpc_decode_kernel" in k:
            continue
        short = k.split("(")[0].split("::")[-1][:40]
        for c, v in cs.items():
            other[short][c] += v
    if name != "sq":
        print("## pass %s: other kernels, summed over all launches of the run" % name)
        for k in sorted(other):
            print("%-42s %s" % (k, " ".join("%s=%.6g" % (c, v) for c, v in sorted(other[k].items()))))
